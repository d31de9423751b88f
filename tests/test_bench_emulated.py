"""bench.py's control flow and JSON contract, dry-run on the CPU: the same main() with torch.cuda stubbed out
(device memory = host memory, streams = no-ops) and phant_amd.Context bound to the emulated library
(tests/emu.py), on miniature workloads.  Checks what the GPU box would only tell at round end: every workload
builds, the timed statuses are the constructed ones, the verdict bookkeeping of the in-flight slots adds up, the
line carries `roofline` and `cpu_baseline`.  The numbers themselves mean nothing here."""
import contextlib
import io
import json
import os
import sys

import pytest

from tests import emu, suite

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

@pytest.fixture(scope="module", autouse=True)
def _emulated_backend():
    # (these runs are about bench.py's plumbing: the node-per-half-wave hash kernel that small batches take -- 32 emulated lanes
    # and ~400 cross-lane operations per permutation -- makes them four times as long; tests/test_emu_verify.py is where it is tested)
    # (the same for node sets of up to 3 500 nodes, a wave per node: tests/test_emu_nodeset.py is where that kernel is tested)
    os.environ["PHANT_TEST_DIAG"] = "verify_no_coop=1,nodeset_wave_max=0"  # (applied to every emulated Context: tests/diag.py)
    try:
        yield from emu.emulated_backend()
    finally:
        os.environ.pop("PHANT_TEST_DIAG", None)


@contextlib.contextmanager
def _no_cuda():
    import torch
    import phant_amd

    class _Stream:
        cuda_stream = 0

        def wait_event(self, *a, **k):
            pass

        def wait_stream(self, *a, **k):
            pass

        def synchronize(self, *a, **k):
            pass

    real_device = torch.device
    saved = (torch.device, torch.cuda.set_device, torch.cuda.current_stream, torch.cuda.Stream, torch.cuda.stream,
             phant_amd.Context, torch.cuda.Event)

    class _Event:  # wall-clock stand-in for a HIP event (the emulated runtime executes at launch)
        def __init__(self, *a, **k):
            self.t = 0.0

        def synchronize(self):
            pass

        def record(self, *a, **k):
            import time
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    def context(device=None, use_torch_stream=True, verify_nodedup=False, dedup_levels=None):
        mode = ("nodedup" if verify_nodedup else "levels%d" % dedup_levels if dedup_levels is not None else "flat")
        return emu.mirror_context(emu.mirror_lib(), mode)

    class _Device:  # torch.device("cuda", i) -> the CPU; isinstance checks inside torch still see a real device
        def __new__(cls, *a, **k):
            return real_device("cpu")

    torch.device = _Device
    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_stream = lambda *a, **k: _Stream()
    torch.cuda.Stream = lambda *a, **k: _Stream()
    torch.cuda.stream = lambda s: contextlib.nullcontext()
    torch.cuda.Event = _Event
    phant_amd.Context = context
    try:
        yield
    finally:
        (torch.device, torch.cuda.set_device, torch.cuda.current_stream, torch.cuda.Stream, torch.cuda.stream,
         phant_amd.Context, torch.cuda.Event) = saved


def _bench(argv, allow_no_line=False):
    sys.path.insert(0, ROOT)
    import bench
    old = sys.argv
    sys.argv = ["bench.py", *argv]
    out = io.StringIO()
    try:
        with _no_cuda(), contextlib.redirect_stdout(out):
            bench.main()
    finally:
        sys.argv = old
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    if allow_no_line and not lines:  # (a rank other than 0)
        return None
    assert len(lines) == 1, out.getvalue()
    return json.loads(lines[0])


def _check_contract(line, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_pass", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == steps and line["warmup"] == warmup
    # ms_per_step is what its name says: steps x ms_per_step = the timed region; a step is `passes_per_timed_step` passes
    c = line["config"]
    assert abs(line["ms_per_step"] * steps - c["timed_region_ms"]) <= 1e-9 * max(1.0, c["timed_region_ms"])
    assert abs(line["ms_per_pass"] * c["passes_per_timed_step"] - line["ms_per_step"]) <= 1e-9 * max(1.0, line["ms_per_step"])
    assert abs(line["value"] - c["units_per_gpu_per_step"] / (line["ms_per_pass"] * 1e-3)) <= 1e-6 * line["value"]
    assert line["vs_baseline"] is None and line["higher_is_better"] is True and "workload" in line["config"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = line["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and "sample" in c


@pytest.mark.parametrize("mode,streams,inner", [("flat", 3, 2), ("flat", 1, 1), ("nodedup", 2, 1)] if suite.FULL else
                         [("flat", 3, 2), ("nodedup", 2, 1)])
def test_config3_dry_run(mode, streams, inner):
    line = _bench(["--proofs", "300", "--steps", "3", "--warmup", "1", "--verify-mode", mode, "--streams",
                   str(streams), "--cpu-seconds", "0.2", "--inner", str(inner), "--block-scale", "0.01"])
    _check_contract(line, 3, 1)
    assert line["metric"] == "mpt_proofs_verified_per_sec_depth8" and line["unit"] == "proofs/s"
    assert line["scaling"] == "weak" and line["config"]["streams"] == streams
    assert line["config"]["passes_per_timed_step"] == inner
    assert line["cpu_baseline"]["statuses_match_gpu_expected"] is True
    # the checker's statuses against what the TIMED launches wrote, over the whole batch
    cb = line["cpu_baseline"]
    assert cb["oracle_checked"] is True and cb["oracle_matches_timed_gpu_statuses"] is True and cb["oracle_checked_proofs"] == 300
    assert line["single_stream"]["value"] > 0
    assert 0 < line["roofline"]["nodes_hashed"] <= line["roofline"]["nodes_shipped"] == 300 * 8
    # BASELINE config 4 (one block witness, strong scaling) rides on the config-3 line
    st = line["strong"]
    assert st["scaling"] == "strong" and st["value"] > 0 and st["proofs_on_rank0"] == 1080
    assert 0 < st["nodes_hashed"] <= st["nodes_shipped"]
    # the predicted ceiling a SCALE reader should hold `value` against sits next to it
    pr = st["predicted"]
    assert pr["n_gpus"] == 1 and pr["ceiling_proofs_per_s"] > 0 and abs(pr["value_over_ceiling"] - st["value"] / pr["ceiling_proofs_per_s"]) < 1e-9


def test_config3_forced_tier_split_dry_run():
    line = _bench(["--proofs", "200", "--steps", "1", "--warmup", "0", "--streams", "1", "--cpu-seconds", "0.2", "--inner", "1",
                   "--dedup-levels", "3", "--no-strong"])
    _check_contract(line, 1, 0)
    assert line["config"]["dedup_levels"] == 3 and "strong" not in line
    # the VALU peak is measured in the run; every kernel of the two-tier pipeline has its own roofline entry (tiers serialised)
    r = line["roofline"]
    assert r["valu"]["peak"] > 0 and "this run" in r["valu"]["peak_source"]
    k = r["kernels"]
    assert k["form"] == "two_tiers" and r["valu_frac"] == r["valu"]["frac"] and 0 < r["valu_bound_frac_of_hbm"]
    assert k["dedup_levels"] == 3 and all(k[n]["ms"] > 0 for n in ("propose_kernel", "hash_deep_kernel", "dedup_kernel",
                                                                    "hash_list_kernel", "walk_kernel"))
    assert k["hash_deep_kernel"]["bound"] == "valu" and k["dedup_kernel"]["bound"] == "hbm"
    assert k["hash_deep_kernel"]["keccak_f"] + k["hash_list_kernel"]["keccak_f"] == r["keccak_f_run"]
    assert "derived" in line["config"]["parity_basis"]
    # (no committed PMC measurement is of a 200-proof batch: nothing is quoted)
    assert r["traffic"] is None


def test_config3_fewer_steps_than_slots():
    line = _bench(["--proofs", "200", "--steps", "1", "--warmup", "0", "--streams", "4", "--cpu-seconds", "0.2", "--inner", "1",
                   "--no-strong"])
    _check_contract(line, 1, 0)


def test_config4_dry_run():
    line = _bench(["--workload", "config4", "--block-scale", "0.01", "--steps", "2", "--warmup", "1", "--streams",
                   "2", "--cpu-seconds", "0.2", "--inner", "1"])
    _check_contract(line, 2, 1)
    assert line["metric"] == "mpt_proofs_verified_per_sec_block_witness" and line["scaling"] == "strong"
    assert line["cpu_baseline"]["statuses_match_gpu_expected"] is True
    assert line["config"]["units_per_gpu_per_step"] == 1080


def test_config2_dry_run():
    line = _bench(["--workload", "config2", "--messages", "3000", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2", "--inner", "2"])
    _check_contract(line, 2, 1)
    assert line["metric"] == "keccak256_136B_hashes_per_sec" and line["config"]["units_per_gpu_per_step"] == 3000


def test_nodeset_and_config5_dry_run():
    line = _bench(["--workload", "nodeset", "--proofs", "300", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2", "--inner", "1"])
    _check_contract(line, 2, 1)
    assert line["cpu_baseline"]["oracle_matches_timed_gpu_statuses"] is True and line["roofline"]["valu"]["frac"] > 0
    assert line["roofline"]["nodes_hashed"] == line["roofline"]["nodes_shipped"] > 300
    # config 5 with every block witness shipped as a node set (phant_mpt_verify_nodeset_submit)
    line = _bench(["--workload", "config5", "--nodeset", "--block-scale", "0.01", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2"])
    _check_contract(line, 2, 1)
    assert line["metric"] == "mpt_proofs_verified_per_sec_block_witness_streamed_nodeset" and "pcie" in line
    assert line["cpu_baseline"]["oracle_matches_timed_gpu_statuses"] is True
    assert line["roofline"]["nodes_hashed"] == line["roofline"]["nodes_shipped"]
    line = _bench(["--workload", "config5", "--stream-proofs", "150", "--steps", "5", "--warmup", "1",
                   "--cpu-seconds", "0.2"])
    _check_contract(line, 5, 1)
    assert "pcie" in line and line["metric"] == "mpt_proofs_verified_per_sec_depth8_streamed"
    # the default: block witnesses (config 4's shape) in rotation
    line = _bench(["--workload", "config5", "--block-scale", "0.01", "--steps", "2", "--warmup", "1", "--cpu-seconds", "0.2"])
    _check_contract(line, 2, 1)
    assert line["metric"] == "mpt_proofs_verified_per_sec_block_witness_streamed" and "pcie" in line
    assert line["config"]["units_per_gpu_per_step"] == 200 + 15 * 8 + 4 * 40 + 1 * 600


def test_block_roots_dry_run():
    line = _bench(["--workload", "block_roots", "--items", "20", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0.2"])
    _check_contract(line, 1, 1)
    assert line["metric"] == "mpt_block_index_roots_per_sec" and line["config"]["units_per_gpu_per_step"] == 3
    assert line["cpu_baseline"]["cores"] == 1 and line["cpu_baseline"]["ms_per_call_of_three"] > 0


def test_mptize_dry_run():
    line = _bench(["--workload", "mptize", "--keys", "3000", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0.2"])
    _check_contract(line, 1, 1)
    assert line["metric"] == "mpt_trie_keys_hashed_per_sec" and line["config"]["units_per_gpu_per_step"] == 3000
    assert line["cpu_baseline"]["root_matches_gpu"] is True
    assert line["roofline"]["algorithmic_bytes_per_launch"] == 3000 * (32 + 78) + 32


def _rank_main(rank, world, port, argv, q):
    """One rank of `torchrun ... bench.py --gpus N`: the environment torchrun would set, RCCL swapped for gloo."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import torch.distributed as dist
    backend = emu.emulated_backend(emu.load_mirror_lib())
    next(backend)
    real_init = dist.init_process_group

    def init(backend_name, device_id=None, **kw):
        assert backend_name == "nccl"  # what bench.py asks for on the GPU box
        return real_init("gloo", **kw)

    dist.init_process_group = init
    try:
        q.put((rank, _bench(argv)))
    except AssertionError as e:  # ranks other than 0 print nothing
        q.put((rank, str(e)))
    finally:
        backend.close()


_TWO_RANK_RUNS = {
    "config3": ["--gpus", "2", "--proofs", "200", "--steps", "2", "--warmup", "1", "--streams", "3", "--inner", "2", "--block-scale", "0.01"],
    "config3-verdicts-every-2-passes": ["--gpus", "2", "--proofs", "200", "--steps", "2", "--warmup", "1", "--streams", "2", "--inner", "3",
                                        "--allreduce-every", "2", "--block-scale", "0.01"],
    "config3-fewer-steps-than-slots": ["--gpus", "2", "--proofs", "200", "--steps", "1", "--warmup", "0", "--streams", "4", "--inner", "1",
                                       "--no-strong"],
    "config4": ["--gpus", "2", "--workload", "config4", "--block-scale", "0.01", "--steps", "2", "--warmup", "1", "--streams", "2",
                "--inner", "1"],
}
_TWO_RANK_IDS = list(_TWO_RANK_RUNS) if suite.FULL else ["config3-verdicts-every-2-passes", "config4"]


@pytest.mark.parametrize("argv", [_TWO_RANK_RUNS[k] for k in _TWO_RANK_IDS], ids=_TWO_RANK_IDS)
def test_two_ranks_dry_run(argv):
    """The N > 1 path the driver launches with torchrun (never run on real GPUs this round): both ranks build their
    shard, agree on the state root, verify, all-reduce the verdict; rank 0 prints the one line."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rank_main, args=(r, 2, port, argv, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    line = got[0]
    assert isinstance(line, dict), line
    assert isinstance(got[1], str)  # rank 1 printed no JSON line (that is what its assertion message says)
    assert line["n_gpus"] == 2 and "cpu_baseline" not in line and line["roofline"]["frac"] > 0
    assert line["config"]["parallelism"] == "key-sharded x2"
    # who took part, gathered over the backend the verdicts take; the verdicts of K passes go as one all-reduce
    assert line["rccl_world"]["ranks"] == 2 and len(line["rccl_world"]["device_ids"]) == 2
    ex = line["roofline"]["verdict_exchange"]
    inner = int(argv[argv.index("--inner") + 1])
    # a verdict per witness by default (rounds 1-3 batched `inner` passes per exchange)
    assert ex["passes_per_allreduce"] == (min(2, inner) if "--allreduce-every" in argv else 1) and ex["allreduces_on_this_rank"] > 0
    if "config4" in argv:
        assert line["scaling"] == "strong"
    elif "--no-strong" not in argv:
        # what a SCALE run reads: the strong-scaling config-4 figure next to the weak config-3 one
        assert line["strong"]["scaling"] == "strong" and line["strong"]["value"] > 0
        # ... with the ceiling two GPUs can reach on ONE block witness next to it
        assert line["strong"]["predicted"]["n_gpus"] == 2 and line["strong"]["predicted"]["ceiling_proofs_per_s"] > 0


def test_comm_form_dry_run():
    """bench.py --comm: ONE process, two (emulated) devices behind one phant_comm -- the shards built per device with the
    shared state root summed in process, a pass = a verify per device + one phant_comm_allreduce_verdict."""
    os.environ["HIPEMU_DEVICES"] = "2"
    try:
        line = _bench(["--comm", "--comm-devices", "2", "--block-scale", "0.01", "--steps", "2", "--warmup", "1", "--inner", "2"])
    finally:
        os.environ.pop("HIPEMU_DEVICES", None)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["parallelism"] == "phant_comm x2 (single process)" and len(line["devices"]) == 2
    assert line["expected"]["speedup_upper_bound_at_8_gpus"]["four_in_flight"] < 8
    line = _bench(["--comm", "--comm-devices", "1", "--block-scale", "0.01", "--steps", "1", "--warmup", "0", "--inner", "1"])
    assert line["n_gpus"] == 1 and line["config"]["units_per_step"] == 1080


def test_smoke_dry_run(capsys):
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench -- on the emulator."""
    import torch
    sys.path.insert(0, ROOT)
    import __graft_entry__ as G
    saved = (torch.cuda.is_available, torch.cuda.current_device)
    torch.cuda.is_available, torch.cuda.current_device = (lambda: True), (lambda: 0)
    try:
        with _no_cuda():
            G.smoke()
    finally:
        torch.cuda.is_available, torch.cuda.current_device = saved
    assert "smoke OK" in capsys.readouterr().out


# ---------------------------------------------------------------- N > 1 without the hardware
_RANK_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from tests import emu
import tests.test_bench_emulated as T
g = emu.emulated_backend()
next(g)
line = None
try:
    line = T._bench({argv!r}, allow_no_line=True)
finally:
    try:
        next(g)
    except StopIteration:
        pass
if line is not None:
    print("LINE " + json.dumps(line), flush=True)
"""


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_a_rank_that_never_shows_up_ends_in_an_error_line_not_a_hang():
    """WORLD_SIZE=2 with only rank 0 started: the rendezvous times out, rank 0 prints a JSON line with "error" and exits non-zero."""
    import subprocess
    port = _free_port()
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PHANT_BENCH_BACKEND="gloo")
    argv = ["--gpus", "2", "--proofs", "100", "--steps", "1", "--warmup", "0", "--rendezvous-seconds", "5", "--max-seconds", "60",
            "--no-cpu-baseline", "--no-strong"]
    p = subprocess.run([sys.executable, "-c", _RANK_SCRIPT.format(root=ROOT, argv=argv)], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert p.returncode in (3, 4), (p.returncode, p.stderr[-2000:])
    err_lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(err_lines) == 1 and err_lines[0]["error"] and err_lines[0]["value"] is None and err_lines[0]["n_gpus"] == 2


_ENTRY_SCRIPT = r"""
import json, os, sys
sys.path.insert(0, {root!r})
from tests import emu
import tests.test_bench_emulated as T
g = emu.emulated_backend()
next(g)
line = None
try:
    line = T._bench(sys.argv[1:], allow_no_line=True)
finally:
    try:
        next(g)
    except StopIteration:
        pass
if line is not None:
    print(json.dumps(line), flush=True)
"""


def test_gpus_2_without_torchrun_relaunches_itself(tmp_path):
    """`python bench.py --gpus 2` the way the driver starts the N = 1 run (no WORLD_SIZE): bench.py becomes the launcher --
    torch.distributed.run, one rank per GPU -- and the job prints ONE line with two ranks in it (gloo and the emulated
    library stand in for RCCL and the GPUs through the test-only PHANT_BENCH_ENTRY script)."""
    import subprocess
    entry = tmp_path / "bench_entry.py"
    entry.write_text(_ENTRY_SCRIPT.format(root=ROOT))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PHANT_BENCH_BACKEND="gloo", PHANT_BENCH_ENTRY=str(entry), PHANT_TEST_DIAG="verify_no_coop=1,nodeset_wave_max=0")
    argv = ["--gpus", "2", "--proofs", "200", "--steps", "1", "--warmup", "0", "--inner", "1", "--no-strong", "--no-cpu-baseline",
            "--max-seconds", "600"]
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, cwd=ROOT, capture_output=True, text=True,
                       timeout=900)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    lines = [json.loads(ln) for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert lines[0]["n_gpus"] == 2 and lines[0]["rccl_world"]["ranks"] == 2 and lines[0]["value"] > 0
