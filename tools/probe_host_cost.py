#!/usr/bin/env python3
"""Cost of ONE verify call as a caller sees it, small batches (the witness of an ordinary block): the device form (arrays
resident: launches + kernels), and the host form (`phant_mpt_verify_batch`: the caller's arrays copied over, statuses and value
locations copied back, through the Python mirror).  Prints host issue time and wall time per call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import phant_amd
from phant_amd import mpt as M

dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)  # (torch's stream: the witness generator's tensor operations and the ctx's launches are ordered)
for n, depth in ((150, 8), (1000, 8), (5000, 8), (100_000, 8)):
    w = phant_amd.witness.account_witness(n, depth=depth, seed=2, device=dev, ctx=ctx)
    b = w.batch
    st = torch.empty(b.n, dtype=torch.uint8, device=dev)
    fc = torch.zeros(1, dtype=torch.int32, device=dev)
    for _ in range(20):
        M.verify_batch_dev(b, status=st, ctx=ctx, fail_count=fc)
    torch.cuda.synchronize()
    N = 2000 if n <= 5000 else 300
    t0 = time.perf_counter()
    for _ in range(N):
        M.verify_batch_dev(b, status=st, ctx=ctx, fail_count=fc)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    # the host form: numpy arrays in, numpy out, one call at a time (each call ends with the results on the host)
    roots = b.roots.cpu().numpy().reshape(-1)
    keys = b.keys.cpu().numpy().reshape(-1)
    nodes = b.nodes.cpu().numpy()
    node_off = b.node_off.cpu().numpy().astype(np.uint64)
    pfn = b.proof_first_node.cpu().numpy().astype(np.uint32)
    for _ in range(5):
        got = M.verify_batch(roots, None, keys, 32, nodes, node_off, pfn, ctx=ctx)
    assert np.array_equal(got[0], w.expected.cpu().numpy())
    Nh = 300 if n <= 5000 else 20
    t3 = time.perf_counter()
    for _ in range(Nh):
        M.verify_batch(roots, None, keys, 32, nodes, node_off, pfn, ctx=ctx)
    t4 = time.perf_counter()
    # ... and the C entry point alone, as a compiled host calls it (pointers prepared once: the Python mirror's array checks and
    # ctypes conversions cost more than the call for small batches)
    status = np.zeros(n, np.uint8); voff = np.zeros(n, np.uint64); vlen = np.zeros(n, np.uint32)
    P = M._np_ptr
    cargs = (ctx.handle, P(roots), 1, None, P(keys), 32, P(nodes), nodes.size, P(node_off), P(pfn), n, P(status), P(voff), P(vlen))
    f = ctx._lib.phant_mpt_verify_batch
    for _ in range(5):
        assert f(*cargs) == 0
    assert np.array_equal(status, w.expected.cpu().numpy())
    t5 = time.perf_counter()
    for _ in range(Nh):
        f(*cargs)
    t6 = time.perf_counter()
    print(f"n={n} ({nodes.size / 1e6:.2f} MB of nodes): device form host issue {1e6 * (t1 - t0) / N:.1f} us, back to back "
          f"{1e6 * (t2 - t0) / N:.1f} us per call; host form (copies in and out) {1e6 * (t6 - t5) / Nh:.1f} us per call of the C entry "
          f"point, {1e6 * (t4 - t3) / Nh:.1f} us through the Python mirror")
