"""BASELINE config 5 on its own workload: consecutive block witnesses (config 4's shape: 80 000 account + storage
proofs against 2 001 roots, a fresh seed per block, 1 % damaged / exclusion) pushed through phant_mpt_verify_submit /
phant_wait from phant_host_alloc (pinned) buffers, three slots in flight, every slot reused several times.  Every
status byte, value range and per-root verdict must equal oracle/verify.c on the same arrays.

The call this stands in for: the witness check a stateless client runs per payload,
/root/reference/src/engine_api/execution_payload.zig:175-181 (a TODO there)."""
import ctypes as C
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BLOCKS = 16   # consecutive witnesses
SLOTS = 3     # in flight


class _Pinned:
    """Arrays in memory from phant_host_alloc (hipHostMalloc), freed with phant_host_free."""

    def __init__(self, ctx):
        self.ctx, self.ptrs = ctx, []

    def array(self, shape, dtype):
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self.ctx.check(self.ctx._lib.phant_host_alloc(self.ctx.handle, max(nbytes, 1), C.byref(p)))
        self.ptrs.append(p)
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def like(self, a):
        out = self.array(a.shape, a.dtype)
        out[...] = a
        return out

    def free(self):
        for p in self.ptrs:
            self.ctx.check(self.ctx._lib.phant_host_free(self.ctx.handle, p))
        self.ptrs = []


def _oracle_sliced(oracle, hb, threads):
    """oracle/verify.c over the whole witness, a slice of proofs per thread (ctypes releases the GIL)."""
    n = hb["pfn"].size - 1
    bounds = [n * t // threads for t in range(threads + 1)]

    def work(t):
        lo, hi = bounds[t], bounds[t + 1]
        pfn = hb["pfn"][lo:hi + 1]
        f0 = int(pfn[0])
        off = hb["node_off"][f0:int(pfn[-1]) + 1]
        st, vo, vl = oracle.mpt_verify_batch(hb["roots"], hb["root_idx"][lo:hi], hb["keys"][lo:hi], 32, hb["nodes"], off,
                                             (pfn - f0).astype(np.uint32))
        return st, vo, vl

    with ThreadPoolExecutor(threads) as ex:
        parts = list(ex.map(work, range(threads)))
    return tuple(np.concatenate([p[k] for p in parts]) for k in range(3))


def test_streamed_block_witnesses_vs_oracle(oracle):
    import phant_amd
    from phant_amd import mpt as M

    ctx = phant_amd.Context()
    pin = _Pinned(ctx)
    threads = max(1, min(os.cpu_count() or 1, 32))
    try:
        blocks, want = [], []
        for k in range(BLOCKS):
            w = phant_amd.witness.block_witness(scale=1.0, seed=500 + k, corrupt_frac=0.01, ctx=ctx)
            b = w.batch
            hb = {"roots": b.roots.cpu().numpy().reshape(-1), "root_idx": b.root_idx.cpu().numpy().astype(np.uint32),
                  "keys": b.keys.cpu().numpy(), "nodes": b.nodes.cpu().numpy(),
                  "node_off": b.node_off.cpu().numpy().astype(np.uint64),
                  "pfn": b.proof_first_node.cpu().numpy().astype(np.uint32)}
            assert hb["pfn"].size - 1 >= 79_000 and hb["roots"].size // 32 == 2001
            st, vo, vl = _oracle_sliced(oracle, hb, threads)
            # (the generator's expectation is a second opinion on the oracle, not the checker)
            assert np.array_equal(st, w.expected.cpu().numpy())
            n_roots = hb["roots"].size // 32
            bad = ~((st == M.PROOF_PRESENT) | (st == M.PROOF_ABSENT))
            verdict = np.bincount(hb["root_idx"][bad], minlength=n_roots).astype(np.uint32)
            want.append((st, vo, vl, verdict))
            n = st.size
            blocks.append({"in": {k2: pin.like(v) for k2, v in hb.items()}, "n": n, "n_roots": n_roots,
                           "status": pin.array((n,), np.uint8), "voff": pin.array((n,), np.uint64),
                           "vlen": pin.array((n,), np.uint32)})
            del w, b
        torch.cuda.synchronize()

        def submit(k, slot):
            x = blocks[k]
            i = x["in"]
            x["status"][:] = 0x55
            ctx.check(ctx._lib.phant_mpt_verify_submit(
                ctx.handle, slot, i["roots"].ctypes.data, x["n_roots"], i["root_idx"].ctypes.data, i["keys"].ctypes.data, 32,
                i["nodes"].ctypes.data, i["nodes"].size, i["node_off"].ctypes.data, i["pfn"].ctypes.data, x["n"],
                x["status"].ctypes.data, x["voff"].ctypes.data, x["vlen"].ctypes.data))

        for round_ in range(2):  # (the second round meets warm arenas: every slot has been used five times by then)
            pending = []
            for k in range(BLOCKS):
                slot = k % SLOTS
                if len(pending) == SLOTS:
                    ctx.check(ctx._lib.phant_wait(ctx.handle, pending.pop(0)))
                submit(k, slot)
                pending.append(slot)
            for s in pending:
                ctx.check(ctx._lib.phant_wait(ctx.handle, s))
            for k in range(BLOCKS):
                x = blocks[k]
                st, vo, vl, verdict = want[k]
                assert np.array_equal(x["status"], st), (k, np.flatnonzero(x["status"] != st)[:10])
                present = st == M.PROOF_PRESENT
                assert np.array_equal(x["voff"][present], vo[present]) and np.array_equal(x["vlen"][present], vl[present])
                got_bad = ~((x["status"] == M.PROOF_PRESENT) | (x["status"] == M.PROOF_ABSENT))
                assert np.array_equal(np.bincount(x["in"]["root_idx"][got_bad], minlength=x["n_roots"]).astype(np.uint32), verdict)
                assert verdict.sum() > 0 and (st == M.PROOF_ABSENT).sum() > 0

        # the per-root verdict as the device forms it (the pipeline's last kernel), on the same witnesses resident
        for k in (0, BLOCKS - 1):
            i = blocks[k]["in"]
            dev = torch.device("cuda", torch.cuda.current_device())
            tb = M.ProofBatch(roots=torch.from_numpy(i["roots"].reshape(-1, 32).copy()).to(dev),
                              root_idx=torch.from_numpy(i["root_idx"].astype(np.int32)).to(dev),
                              keys=torch.from_numpy(i["keys"].copy()).to(dev), nodes=torch.from_numpy(i["nodes"].copy()).to(dev),
                              node_off=torch.from_numpy(i["node_off"].astype(np.int64)).to(dev),
                              proof_first_node=torch.from_numpy(i["pfn"].astype(np.int32)).to(dev))
            fails = torch.empty(blocks[k]["n_roots"], dtype=torch.int32, device=dev)
            st_dev = M.verify_batch_dev(tb, ctx=ctx, fail_count=fails)
            torch.cuda.synchronize()
            assert np.array_equal(st_dev.cpu().numpy(), want[k][0])
            assert np.array_equal(fails.cpu().numpy().astype(np.uint32), want[k][3])
    finally:
        pin.free()
        ctx.close()


def test_streamed_block_witnesses_as_node_sets_vs_oracle(oracle):
    """The same stream with every block witness shipped as a node SET -- the form the hook at
    /root/reference/src/engine_api/execution_payload.zig:121,175-181 would be handed (`executionWitness`: every trie node once,
    in any order): 16 consecutive witnesses of 80 000 keys against 2 001 roots through phant_mpt_verify_nodeset_submit /
    phant_wait, three slots in flight, every slot reused; every status byte, value range and per-root verdict must equal
    oracle/verify.c's node-set verifier on the same arrays (and the generator's expectation: a damaged copy only costs its
    key the proof when no intact copy of that node came with another key)."""
    import phant_amd
    from phant_amd import mpt as M

    ctx = phant_amd.Context()
    pin = _Pinned(ctx)
    threads = max(1, min(os.cpu_count() or 1, 16))
    try:
        blocks, want, shipped = [], [], []
        for k in range(BLOCKS):
            w = phant_amd.witness.block_witness(scale=1.0, seed=700 + k, corrupt_frac=0.01, ctx=ctx)
            s = phant_amd.witness.node_set(w, ctx=ctx, shuffle_seed=k)
            hb = {"roots": s.roots.cpu().numpy().reshape(-1), "root_idx": s.root_idx.cpu().numpy().astype(np.uint32),
                  "keys": s.keys.cpu().numpy(), "nodes": s.nodes.cpu().numpy(),
                  "node_off": s.node_off.cpu().numpy().astype(np.uint64)}
            n, n_roots = hb["keys"].shape[0], hb["roots"].size // 32
            assert n >= 79_000 and n_roots == 2001
            shipped.append((int(w.batch.nodes.numel()), hb["nodes"].size))
            expected = w.expected_nodeset.cpu().numpy()
            del w, s
            blocks.append({"in": {k2: pin.like(v) for k2, v in hb.items()}, "n": n, "n_roots": n_roots, "expected": expected,
                           "status": pin.array((n,), np.uint8), "voff": pin.array((n,), np.uint64),
                           "vlen": pin.array((n,), np.uint32)})
        torch.cuda.synchronize()

        # the oracle hashes and orders a block's ~300 000 nodes single-threaded: the blocks side by side
        def check(k):
            i = blocks[k]["in"]
            return oracle.mpt_verify_nodeset(i["roots"], i["root_idx"], i["keys"], 32, i["nodes"], i["node_off"])

        with ThreadPoolExecutor(threads) as ex:
            want = list(ex.map(check, range(BLOCKS)))
        for k in range(BLOCKS):
            assert np.array_equal(want[k][0], blocks[k]["expected"])  # (the generator: a second opinion on the oracle)

        def submit(k, slot):
            x = blocks[k]
            i = x["in"]
            x["status"][:] = 0x55
            ctx.check(ctx._lib.phant_mpt_verify_nodeset_submit(
                ctx.handle, slot, i["roots"].ctypes.data, x["n_roots"], i["root_idx"].ctypes.data, i["keys"].ctypes.data, 32,
                i["nodes"].ctypes.data, i["nodes"].size, i["node_off"].ctypes.data, i["node_off"].size - 1, x["n"],
                x["status"].ctypes.data, x["voff"].ctypes.data, x["vlen"].ctypes.data))

        for round_ in range(2):
            pending = []
            for k in range(BLOCKS):
                slot = k % SLOTS
                if len(pending) == SLOTS:
                    ctx.check(ctx._lib.phant_wait(ctx.handle, pending.pop(0)))
                submit(k, slot)
                pending.append(slot)
            for s_ in pending:
                ctx.check(ctx._lib.phant_wait(ctx.handle, s_))
            for k in range(BLOCKS):
                x = blocks[k]
                st, vo, vl = want[k]
                assert np.array_equal(x["status"], st), (k, np.flatnonzero(x["status"] != st)[:10])
                present = st == M.PROOF_PRESENT
                assert np.array_equal(x["voff"][present], vo[present]) and np.array_equal(x["vlen"][present], vl[present])
                assert (st == M.PROOF_MISSING_NODE).sum() > 0 and (st == M.PROOF_ABSENT).sum() > 0
        # a node set is what crosses the bus: well under the per-proof form's bytes
        assert all(ns < 0.7 * pp for pp, ns in shipped), shipped[:3]

        # the per-root verdict as the device form computes it (the walk kernel), on the same witnesses resident
        for k in (0, BLOCKS - 1):
            i = blocks[k]["in"]
            dev = torch.device("cuda", torch.cuda.current_device())
            fails = torch.empty(blocks[k]["n_roots"], dtype=torch.int32, device=dev)
            st_dev = M.verify_nodeset_dev(torch.from_numpy(i["roots"].reshape(-1, 32).copy()).to(dev),
                                          torch.from_numpy(i["root_idx"].astype(np.int32)).to(dev),
                                          torch.from_numpy(i["keys"].copy()).to(dev), torch.from_numpy(i["nodes"].copy()).to(dev),
                                          torch.from_numpy(i["node_off"].astype(np.int64)).to(dev), ctx=ctx, fail_count=fails)
            torch.cuda.synchronize()
            st = want[k][0]
            assert np.array_equal(st_dev.cpu().numpy(), st)
            bad = ~((st == M.PROOF_PRESENT) | (st == M.PROOF_ABSENT))
            assert np.array_equal(fails.cpu().numpy().astype(np.uint32),
                                  np.bincount(i["root_idx"][bad], minlength=blocks[k]["n_roots"]).astype(np.uint32))
    finally:
        pin.free()
        ctx.close()
