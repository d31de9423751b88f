import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes as C
import phant_amd
from phant_amd import mpt as M
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
w = phant_amd.witness.account_witness(100000, depth=8, seed=2, device=dev, ctx=ctx, corrupt_frac=0.0)
s = phant_amd.witness.node_set(w, ctx=ctx)
st = torch.empty(s.n, dtype=torch.uint8, device=dev)
lib = ctx._lib
lib.phant_debug_ns_header.restype = C.c_int32
lib.phant_debug_ns_header.argtypes = [C.c_void_p, C.c_void_p]
for nk in (64, 1000, 100000):
    keys = s.keys[:nk].contiguous()
    for _ in range(3):
        M.verify_nodeset_dev(s.roots, None, keys, s.nodes, s.node_off, status=st[:nk], ctx=ctx)
    torch.cuda.synchronize()
    hdr = (C.c_uint32 * 2048)()
    ctx.check(lib.phant_debug_ns_header(ctx.handle, hdr))
    n = hdr[699]
    t = [hdr[700 + k] for k in range(n + 1)]
    d = [(t[k + 1] - t[k]) & 0xffffffff for k in range(n)]
    print(nk, "keys;", n, "marks; cycles:", d, "sum", sum(d), flush=True)
