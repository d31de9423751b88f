"""TEST INFRASTRUCTURE: the library reads no environment variable -- its measured alternatives and test hooks are per-ctx switches
(include/phant_gpu_diag.h: phant_diag_set).  Tests that want a whole test module to run under a switch (a child pytest per
setting) name it in PHANT_TEST_DIAG="knob=value,knob=value" and every Context the tests create gets it applied here."""
import os


def settings():
    spec = os.environ.get("PHANT_TEST_DIAG", "").strip()
    out = []
    for item in filter(None, (x.strip() for x in spec.split(","))):
        k, _, v = item.partition("=")
        out.append((k.strip(), int(v or "1")))
    return out


def apply(ctx):
    for k, v in settings():
        ctx.diag_set(k, v)
    return ctx


def install():
    """phant_amd.Context(...) applies PHANT_TEST_DIAG from now on (idempotent)."""
    from phant_amd import context as Cx
    if getattr(Cx.Context, "_test_diag_installed", False):
        return
    orig = Cx.Context.__init__

    def init(self, *a, **k):
        orig(self, *a, **k)
        apply(self)

    Cx.Context.__init__ = init
    Cx.Context._test_diag_installed = True
