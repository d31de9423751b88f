#!/usr/bin/env python3
"""bench.py -- BASELINE.json's headline metric on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--proofs P] [--workload config3|config2|config4|config5|nodeset]

A "step" = one pass of the hot path over one batch of synthetic input already
resident in HBM:
  config3 (default, the configuration the metric is quoted on): verify
      P = 100 000 synthetic depth-8 account proofs against one state root
      (7 full 532-byte branch nodes + one 112-byte account leaf per proof, each
      proof shipped as its own node list, 0.5 % corrupted + 0.5 % exclusion
      proofs), reduce to one pass/fail word per root.
  config2: Keccak-256 of 1 048 576 x 136-byte messages (sponge kernel only).
  nodeset: config 3's proofs as a node SET (every distinct node shipped once, references resolved by
      hash; phant_mpt_verify_nodeset_dev).
  config4: the witness of one 10 000-transaction block (phant_amd.witness.BLOCK_10K_TX: 20 000 depth-8 account
      proofs + 60 000 storage proofs over 2 000 contracts of depth 3 / 5 / 7, 2 001 roots) as one multi-root
      batch, sharded over the N GPUs (accounts by top key nibble, contracts whole): STRONG scaling, one
      all-reduce of the per-root failure counts per step.
  mptize: the state-trie hasher (src/mpt/mpt.zig:38-119 mptize): the root of the trie of --keys sorted random 32-byte
      keys with 78-byte values (account bodies), arrays resident in HBM (phant_mpt_root_dev).
  config5: consecutive block witnesses streamed from pinned host memory through
      phant_mpt_verify_submit / phant_wait (copy-in of witness k+1 overlaps the kernels of witness k);
      a step = one block witness (config 4's shape; --stream-proofs P: account witnesses of P depth-8 proofs instead);
      --steps 64 = 256 witnesses (4 back-to-back per step); PCIe-bound by construction.

N > 1: one process per GPU (torchrun), proofs sharded by the top key nibble,
every rank verifies its own P proofs (weak scaling); the only data-path
collective is the all-reduce of the per-root failure count (RCCL).  value =
proofs of ALL ranks / max-over-ranks time.  --streams S (default 2) keeps S independent launch
sequences in flight per GPU (a sequence is two HIP streams -- the tiers run next to each other --, and two of them are the four
hardware queues ROCm gives a process: with more in flight, streams share queues and wait for each other; measured 2 / 3 / 4 / 6:
530 / 484 / 518 / 507 M proofs/s, profiles/EXPERIMENTS.md), each on its own ctx + HIP stream and each over a DIFFERENT witness (other seed: a
validator verifying consecutive witnesses; no step re-reads the bytes the previous step on its slot left in
L2 / Infinity Cache): every step is still a full pass over a full batch; `single_stream` in the JSON line is
the same number of passes strictly one after the other (alternating witnesses), and `roofline.achieved` always
refers to ONE launch.  A timed "step" is repeated --inner times back to back (default 30) so that the timed
region is >= 100 ms; all per-step figures are per single pass.  The config-3 line also carries `strong`: the
config-4 block witness split over the same N GPUs (strong scaling), measured right after.

Prints ONE JSON line (rank 0) with `roofline` (HBM; algorithmic bytes per
launch / average kernel duration measured with HIP events on the launch
stream) and, at N = 1, `cpu_baseline` (the CPU oracle timed on this host).
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def csrc_sha256():
    """The hash tools/pmc_traffic.py stamps its result with: one SHA-256 over the kernel sources (sorted by name)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "phant_amd", "csrc")
    for f in sorted(os.listdir(d)):
        h.update(f.encode() + b"\0" + open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def pmc_traffic(mode, proofs):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in their own runs and
    corrected as tools/pmc_traffic.py documents).  The counters cannot be read from inside this process, so this is a quoted
    file -- and only quoted while it is a measurement of THESE kernels: the newest profiles/*/pmc_traffic.json whose
    `csrc_sha256` equals the hash of phant_amd/csrc as it is now.  Otherwise {"bytes": None, "stale": ...}."""
    import glob
    if proofs != 100_000:
        return None
    now = csrc_sha256()
    newest = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")), key=os.path.getmtime, reverse=True):
        try:
            t = json.load(open(f))
        except (OSError, ValueError):
            continue
        if mode not in t:
            continue
        newest = newest or os.path.relpath(f, ROOT)
        if t.get("csrc_sha256") == now:
            return {"bytes": t[mode]["total"], "read": t[mode]["read"], "write": t[mode]["write"],
                    "source": f"static:{os.path.relpath(f, ROOT)}@csrc:{now[:12]}",
                    "kernels": {k.split("::")[-1]: v["read"] + v["write"] for k, v in t[mode].get("kernels", {}).items()}}
    return {"bytes": None, "stale": f"no profiles/*/pmc_traffic.json was measured on the kernel sources as they are now "
                                    f"(csrc:{now[:12]}); newest: {newest}"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--proofs", type=int, default=100_000, help="proofs per GPU (config3)")
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--proof-order", choices=("random", "sorted"), default="random",
                    help="config 3: order of the proofs in the batch (BASELINE: random; sorted = ascending keys, an A/B)")
    ap.add_argument("--workload", default="config3", choices=["config3", "config2", "config4", "config5", "nodeset", "mptize", "block_roots"])
    ap.add_argument("--items", type=int, default=100, help="block_roots: items per list (transactions / receipts / withdrawals)")
    ap.add_argument("--keys", type=int, default=1_000_000, help="mptize: sorted 32-byte keys (78-byte values) per GPU")
    ap.add_argument("--block-scale", type=float, default=1.0, help="config4: size of the block relative to 10k tx")
    ap.add_argument("--stream-proofs", type=int, default=0,
                    help="config5: stream account witnesses of this many depth-8 proofs instead of block witnesses (A/B)")
    ap.add_argument("--stream-slots", type=int, default=2, help="witnesses in flight (config5)")
    ap.add_argument("--nodeset", action="store_true",
                    help="config5: stream the block witnesses as node SETS (every distinct node once: the form an execution witness has, "
                         "src/engine_api/execution_payload.zig:121) through phant_mpt_verify_nodeset_submit / phant_wait")
    ap.add_argument("--verify-mode", default="flat", choices=["flat", "nodedup"],
                    help="flat = the verify pipeline as the library sizes it (shallow trie levels deduplicated, deep ones hashed "
                         "in place); nodedup = every shipped node hashed (= --dedup-levels 0: A/B)")
    ap.add_argument("--diag", default="", help="per-ctx switches of include/phant_gpu_diag.h for every ctx of the run: "
                                               "knob=value,knob=value (phant_amd.Context.DIAG; e.g. verify_serial=1 for the PMC passes)")
    ap.add_argument("--dedup-levels", type=int, default=None,
                    help="flat: trie levels deduplicated (default: chosen from the batch size)")
    ap.add_argument("--inner", type=int, default=30,
                    help="config3 / config4 / config2 / nodeset: back-to-back passes per timed step (timed region >= 100 ms)")
    ap.add_argument("--no-strong", action="store_true", help="config3: skip the config-4 strong-scaling object")
    ap.add_argument("--streams", type=int, default=2,
                    help="config3: independent batches in flight, each on its own ctx + HIP stream (a validator "
                         "verifying consecutive witnesses); 1 = strictly one launch sequence after the other")
    ap.add_argument("--allreduce-every", type=int, default=1,
                    help="N > 1 (>= 1): a slot's per-root verdicts are exchanged once per this many passes, as one all-reduce of a "
                         "(passes x roots) block on a stream of its own (default 1: a verdict per witness, what a validator needs; "
                         "rounds 1-3 of this repository batched --inner = 30 passes per exchange: their N > 1 figures are not "
                         "comparable)")
    ap.add_argument("--one-device", action="store_true",
                    help="N > 1: every rank on device 0 (a one-GPU box: the process-per-GPU control flow -- shards, per-pass verdict "
                         "all-reduce, the strong leg -- over the real kernels; with PHANT_BENCH_BACKEND=gloo the collectives are staged "
                         "through host memory.  NOT a scaling measurement: the line says distinct_devices = 1)")
    ap.add_argument("--comm", action="store_true",
                    help="the in-process form of the multi-GPU path: ONE process, every visible device behind one phant_comm "
                         "(phant_comm_ctx + phant_mpt_verify_verdict_dev per device, one phant_comm_allreduce_verdict per pass); "
                         "prints config 4's strong-scaling line.  No torchrun")
    ap.add_argument("--comm-devices", type=int, default=0, help="--comm: devices to use (default 0 = all visible)")
    ap.add_argument("--messages", type=int, default=1 << 20, help="config2: 136-byte messages per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="config3 at N = 1: skip the short extra legs (config 2, mptize, config 5 at 8 blocks) that the default run "
                         "appends to the line as `extra` so that the driver's record carries them")
    ap.add_argument("--extra-seconds", type=int, default=420, help="wall-clock budget of all extra legs together")
    ap.add_argument("--max-seconds", type=int, default=1500,
                    help="wall-clock guard: past this the process prints a JSON line with \"error\" (rank 0) and exits with rc 4 "
                         "instead of hanging (a rendezvous or RCCL initialisation that never completes)")
    ap.add_argument("--rendezvous-seconds", type=int, default=180, help="N > 1: timeout of the process-group rendezvous and of every collective")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    args = ap.parse_args()
    if args.allreduce_every < 1:
        ap.error("--allreduce-every must be >= 1 (1 = a verdict per pass; --inner = one exchange per timed step)")
    return args


def cpu_baseline_config3(w, target_seconds, gpu_status=None):
    """The CPU oracle (restatement of the reference's scalar path) on a bounded sample of the same
    workload, 1 core -- phant's MPT code is single-threaded."""
    import numpy as np
    from oracle import oracle as O

    b = w.batch
    n = b.n
    probe = min(n, 2000)

    def run(cnt):
        nn = cnt * w.nodes_per_proof
        nodes = b.nodes[: int(b.node_off[nn].item())].cpu().numpy()
        node_off = b.node_off[: nn + 1].cpu().numpy().astype(np.uint64)
        pfn = b.proof_first_node[: cnt + 1].cpu().numpy().astype(np.uint32)
        keys = b.keys[:cnt].cpu().numpy()
        roots = b.roots.cpu().numpy()
        t0 = time.perf_counter()
        st, _, _ = O.mpt_verify_batch(roots, None, keys, 32, nodes, node_off, pfn)
        dt = time.perf_counter() - t0
        return st, dt

    st, dt = run(probe)
    rate = probe / dt
    cnt = int(max(probe, min(n, rate * target_seconds)))
    st, dt = run(cnt)
    ok = bool((st == w.expected[:cnt].cpu().numpy()).all())
    passes = 1
    if cnt == n:  # the whole batch is less CPU work than asked for: verify it again until the sample is long enough
        while dt < target_seconds and passes < 16:
            st2, dt2 = run(cnt)
            ok = ok and bool((st2 == st).all())
            dt += dt2
            passes += 1
    # the checker's statuses against what the TIMED launches wrote (slot 0's last pass over this witness), not only
    # against the generator's expectation
    gpu_ok = None
    if gpu_status is not None:
        gpu_ok = bool((st == gpu_status[:cnt]).all())
    out = {"value": cnt * passes / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
           "sample": f"first {cnt} of the {n} config-3 proofs{' x %d passes' % passes if passes > 1 else ''}, "
                     f"oracle/verify.c single-threaded, {dt:.1f} s",
           "host_cpus": os.cpu_count(), "statuses_match_gpu_expected": ok, "oracle_checked": gpu_ok is not None,
           "oracle_matches_timed_gpu_statuses": gpu_ok, "oracle_checked_proofs": cnt if gpu_ok is not None else 0}
    # the same scalar code on all host cores, one slice of proofs per thread (ctypes releases the GIL);
    # phant itself is single-threaded, so this is the generous reading of "the CPU path"
    try:
        from concurrent.futures import ThreadPoolExecutor
        threads = max(1, min(os.cpu_count() or 1, 128))
        npp = w.nodes_per_proof
        nodes_h = b.nodes.cpu().numpy()
        off_h = b.node_off.cpu().numpy().astype(np.uint64)
        pfn_h = b.proof_first_node.cpu().numpy().astype(np.uint32)
        keys_h = b.keys.cpu().numpy()
        roots_h = b.roots.cpu().numpy()
        bounds = [n * t // threads for t in range(threads + 1)]
        budget = max(2.0, target_seconds / 2.0)  # wall seconds, whatever the host's scaling turns out to be

        def work(t):
            lo, hi = bounds[t], bounds[t + 1]
            if hi <= lo:
                return 0
            off_t = off_h[lo * npp: hi * npp + 1]
            pfn_t = pfn_h[lo: hi + 1] - pfn_h[lo]
            done = 0
            while time.perf_counter() < deadline:
                O.mpt_verify_batch(roots_h, None, keys_h[lo:hi], 32, nodes_h, off_t, pfn_t)
                done += hi - lo
            return done

        with ThreadPoolExecutor(threads) as ex:
            t0 = time.perf_counter()
            deadline = t0 + budget
            done = sum(ex.map(work, range(threads)))
            dta = time.perf_counter() - t0
        out["all_cores"] = {"value": done / dta, "unit": "proofs/s", "cores": threads,
                            "sample": f"{done} proofs (slices of the {n}, repeated), {threads} threads, {dta:.1f} s"}
    except Exception as exc:  # the 1-core figure above is the baseline; this one is a courtesy
        out["all_cores"] = {"error": repr(exc)}
    return out


def cpu_baseline_block(w, target_seconds):
    """config 4: the CPU oracle, 1 core, on an evenly strided sample of the block's proofs (all depth classes)."""
    import numpy as np
    from oracle import oracle as O
    from phant_amd.shard import HostBatch, take_proofs

    b = w.batch
    hb = HostBatch(roots=b.roots.cpu().numpy(), root_idx=b.root_idx.cpu().numpy().astype(np.uint32),
                   keys=b.keys.cpu().numpy(), nodes=b.nodes.cpu().numpy(),
                   node_off=b.node_off.cpu().numpy().astype(np.uint64),
                   proof_first_node=b.proof_first_node.cpu().numpy().astype(np.uint32))
    exp = w.expected.cpu().numpy()

    def run(cnt):
        idx = np.linspace(0, hb.n - 1, cnt).astype(np.int64)
        sub = take_proofs(hb, idx)
        t0 = time.perf_counter()
        st, _, _ = O.mpt_verify_batch(sub.roots, sub.root_idx, sub.keys, 32, sub.nodes, sub.node_off, sub.proof_first_node)
        return st, time.perf_counter() - t0, idx

    probe = min(hb.n, 2000)
    st, dt, idx = run(probe)
    cnt = int(max(probe, min(hb.n, probe / dt * target_seconds)))
    st, dt, idx = run(cnt)
    return {"value": cnt / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": f"{cnt} evenly strided proofs of the {hb.n}-proof block witness, oracle/verify.c single-threaded, "
                      f"{dt:.1f} s", "host_cpus": os.cpu_count(),
            "statuses_match_gpu_expected": bool((st == exp[idx]).all())}


def cpu_baseline_nodeset(sset, target_seconds, gpu_status=None):
    """oracle/verify.c's node-set verifier (hash every node, order the digests, walk every key), 1 core, over the WHOLE witness --
    it is seconds of CPU work --, repeated until the sample is long enough; its statuses against the timed GPU launches'."""
    import numpy as np
    from oracle import oracle as O

    roots = sset.roots.cpu().numpy()
    ri = None if sset.root_idx is None else sset.root_idx.cpu().numpy().astype(np.uint32)
    keys = sset.keys.cpu().numpy()
    nodes = sset.nodes.cpu().numpy()
    off = sset.node_off.cpu().numpy().astype(np.uint64)
    dt, passes, st = 0.0, 0, None
    while passes == 0 or (dt < target_seconds and passes < 8):
        t0 = time.perf_counter()
        st_k, _, _ = O.mpt_verify_nodeset(roots, ri, keys, 32, nodes, off)
        dt += time.perf_counter() - t0
        passes += 1
        assert st is None or np.array_equal(st, st_k)
        st = st_k
    gpu_ok = None if gpu_status is None else bool(np.array_equal(st, gpu_status))
    return {"value": sset.n * passes / dt, "unit": "proofs/s", "cores": 1, "kind": "port",
            "sample": f"the whole node set ({sset.total_nodes} nodes, {sset.n} keys) x {passes} pass(es), oracle/verify.c "
                      f"(oracle_mpt_verify_nodeset) single-threaded, {dt:.1f} s",
            "host_cpus": os.cpu_count(), "oracle_checked": gpu_ok is not None, "oracle_matches_timed_gpu_statuses": gpu_ok,
            "oracle_checked_proofs": sset.n if gpu_ok is not None else 0}


def cpu_baseline_block_roots(lists, target_seconds):
    """oracle/mpt.c (the restatement of mptize behind calculateMPTRoot, blockchain.zig:209-235), one core, the same three lists."""
    from oracle import oracle as O
    reps, t0 = 0, time.perf_counter()
    while reps == 0 or time.perf_counter() - t0 < target_seconds:
        for x in lists:
            O.index_root_rlp(x)
        reps += 1
    dt = time.perf_counter() - t0
    return {"value": 3 * reps / dt, "unit": "roots/s", "cores": 1, "kind": "port", "ms_per_call_of_three": dt / reps * 1e3,
            "sample": f"{reps} x the same three lists, oracle/mpt.c (oracle_index_root_rlp) single-threaded, {dt:.1f} s",
            "host_cpus": os.cpu_count()}


def cpu_baseline_config2(blob, n, target_seconds):
    import numpy as np
    from oracle import oracle as O

    cnt = min(n, 200_000)
    host = blob[: cnt * 136].cpu().numpy()
    off = np.arange(cnt + 1, dtype=np.uint64) * 136
    t0 = time.perf_counter()
    O.keccak256_batch(host, off)
    dt = time.perf_counter() - t0
    reps = max(1, int(target_seconds / max(dt, 1e-3)))
    t0 = time.perf_counter()
    for _ in range(reps):
        O.keccak256_batch(host, off)
    dt = time.perf_counter() - t0
    return {"value": cnt * reps / dt, "unit": "hashes/s", "cores": 1, "kind": "port",
            "sample": f"{reps} x first {cnt} messages, oracle/keccak.c single-threaded, {dt:.1f} s",
            "host_cpus": os.cpu_count()}


def cpu_baseline_mptize(keys_t, vals_t, n, root_t, target_seconds):
    """oracle/mpt.c (the restatement of mpt.zig:38-119), 1 core, on a prefix of the same sorted keys; at full size the
    root is compared with the GPU's."""
    import numpy as np
    from oracle import oracle as O

    def run(cnt):
        kb = keys_t[: cnt * 32].cpu().numpy()
        vb = vals_t[: cnt * 78].cpu().numpy()
        ko = (np.arange(cnt + 1, dtype=np.uint32) * 32)
        vo = (np.arange(cnt + 1, dtype=np.uint64) * 78)
        t0 = time.perf_counter()
        r = O.mptize_packed(kb, ko, vb, vo)
        return r, time.perf_counter() - t0

    probe = min(n, 20000)
    _, dt = run(probe)
    cnt = int(max(probe, min(n, probe / dt * target_seconds)))
    r, dt = run(cnt)
    out = {"value": cnt / dt, "unit": "keys/s", "cores": 1, "kind": "port",
           "sample": f"first {cnt} of the {n} sorted keys, oracle/mpt.c single-threaded, {dt:.1f} s", "host_cpus": os.cpu_count()}
    if cnt == n:
        out["root_matches_gpu"] = bool(bytes(r) == bytes(root_t.cpu().numpy().tobytes()))
    return out


def trie_keccak_f(keys32, value_len):
    """Keccak-f the trie hasher must run for sorted distinct 32-byte keys with fixed-length values: one per node and rate
    block, from the trie's SHAPE (mpt.zig:47-119): a leaf per key, a branch wherever the keys under a nibble prefix go on
    with two or more different nibbles (its length from the number of children: every child of such a trie is a 32-byte
    reference), an extension where a shared run of nibbles ends in a branch.  Exact while 15 nibbles tell all keys apart."""
    import torch
    n = keys32.shape[0]
    be = torch.zeros(n, dtype=torch.int64, device=keys32.device)
    for t in range(7):  # the first 60 bits (15 nibbles) as a non-negative integer
        be = (be << 8) | keys32[:, t].to(torch.int64)
    be = (be << 4) | (keys32[:, 7].to(torch.int64) >> 4)
    # a leaf: list header + hex-prefix path + value; all of them one block here (the caller's values are short)
    leaf_len = 3 + 33 + 2 + value_len
    perms = n * (leaf_len // 136 + 1)
    nodes = {"leaves": n, "branches": 0, "extensions": 0}
    prev_single = None  # per group of depth d - 1: did it go on with ONE nibble (part of an extension's run)?
    for d in range(0, 15):
        pre = be >> (60 - 4 * d) if d else torch.zeros_like(be)
        nxt = be >> (60 - 4 * (d + 1))
        groups, inv, cnt = torch.unique_consecutive(pre, return_inverse=True, return_counts=True)
        multi = cnt >= 2
        if not bool(multi.any()):
            break
        kids = torch.unique_consecutive(nxt)
        kid_parent = kids >> 4
        _, kid_cnt = torch.unique_consecutive(kid_parent, return_counts=True)  # children per group (all groups, in order)
        c = kid_cnt[multi]
        br = c[c >= 2]
        blen = 32 * br + 17
        blen = blen + torch.where(blen > 255, 3, torch.where(blen > 55, 2, 1))
        perms += int((blen // 136 + 1).sum().item())
        nodes["branches"] += int(br.numel())
        # an extension sits on top of every maximal run of single-child groups: count the runs' heads = single-child
        # groups whose parent group was a branch (or the root)
        single = multi & (kid_cnt == 1)
        if bool(single.any()):
            if d == 0:
                heads = int(single.sum().item())
            else:
                parent = groups >> 4
                ppre, pinv = torch.unique_consecutive(parent, return_inverse=True)
                parent_single = prev_single[pinv] if prev_single is not None else torch.zeros_like(single)
                heads = int((single & ~parent_single).sum().item())
            perms += heads
            nodes["extensions"] += heads
        prev_single = single
    return perms, nodes


def device_id_bytes(dev):
    """16 bytes that identify the physical device behind `dev` (its UUID; PCI bus id + ordinal where the build has none)."""
    import torch
    try:
        p = torch.cuda.get_device_properties(dev)
        u = getattr(p, "uuid", None)
        raw = bytes.fromhex(str(u).replace("-", "")) if u is not None else b""
    except Exception:
        raw = b""
    if len(raw) != 16:
        import hashlib
        raw = hashlib.sha256(f"{os.uname().nodename}:{getattr(dev, 'index', 0)}:{os.environ.get('LOCAL_RANK', '0')}".encode()).digest()[:16]
    return torch.tensor(list(raw), dtype=torch.uint8, device=dev)


# What strong scaling of ONE block witness can reach: at N = 8 a rank holds 10 000 of config 4's 80 000 proofs -- below the
# size where the two tiers pay (72 MB of nodes), so it takes the S = 0 form, whose launch is latency-bound (one wave's
# Keccak-f chain per node), and every pass ends in an all-reduce.  Measured on one MI355X (profiles/r2_d/
# small_batches_direct_walk.jsonl): 10 000 depth-8 proofs 74.6 us per launch (134 M proofs/s), 282 M proofs/s with four
# launches in flight.
STRONG_EXPECTED = {"basis": "one GPU, 10 000 proofs per launch (what a rank of 8 holds): 74.6 us per launch one at a time, "
                            "282 M proofs/s with four in flight (profiles/r2_d/small_batches_direct_walk.jsonl); one GPU, the "
                            "whole 80 000-proof witness: ~0.20 ms per launch, ~515 M proofs/s with four in flight",
                   "speedup_upper_bound_at_8_gpus": {"four_in_flight": round(8 * 282 / 515, 1), "one_at_a_time": round(8 * 134 / 400, 1)},
                   "why": "a rank's share of one block falls under the latency floor of a launch (the S = 0 form); the verdict "
                          "exchange (2 001 x 4 B) is latency-bound as well"}


# one GPU's measured rate (M proofs/s, launches in flight) at the number of block-witness proofs a rank holds: profiles/r2_d/
# small_batches_direct_walk.jsonl, block_witness_scale_vs_levels.jsonl, profiles/r3_final/bench_config4.json
_RATE_POINTS = [(8_000, 276e6), (10_000, 282e6), (16_000, 352e6), (24_000, 368e6), (80_000, 520e6)]


def strong_predicted(total_proofs, world, value):
    """What N GPUs can reach on ONE block witness if nothing but the per-launch latency floor limits a rank: N x the
    single-GPU rate at total / N proofs per launch (piecewise linear between the measured points)."""
    per = total_proofs / world
    pts = _RATE_POINTS
    if per <= pts[0][0]:
        rate = pts[0][1] * per / pts[0][0]
    elif per >= pts[-1][0]:
        rate = pts[-1][1]
    else:
        for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
            if x0 <= per <= x1:
                rate = y0 + (y1 - y0) * (per - x0) / (x1 - x0)
                break
    ceiling = rate * world
    return {"n_gpus": world, "proofs_per_rank": per, "ceiling_proofs_per_s": ceiling, "value_over_ceiling": value / ceiling,
            "note": "ceiling = n_gpus x (one GPU's measured rate at proofs_per_rank per launch); read `value` against THIS, not "
                    "against n_gpus x the N = 1 value: a rank's share of one block falls under the latency floor of a launch"}


def run_comm_bench(args):
    """--comm: ONE process, every visible device behind one phant_comm (what phant itself, a single process, would use).
    Config 4's block witness split over the devices (device-resident shards), a pass = phant_mpt_verify_verdict_dev on
    every device's ctx + one phant_comm_allreduce_verdict; prints the strong-scaling line."""
    import torch
    import phant_amd
    from phant_amd import mpt as M
    from phant_amd.comm import Comm

    D = args.comm_devices or torch.cuda.device_count()
    comm = Comm(n_devices=D)
    D = comm.size
    inner, steps, warmup = max(1, args.inner), args.steps, args.warmup
    # ---- every device builds its shard.  The shared state root needs every rank's level-1 hashes: a first pass over the
    # ranks collects them (its witnesses are thrown away), the second builds the shards against their sum
    slots, contribs = [None] * D, []

    def build(d, share):
        torch.cuda.set_device(d)
        dev = torch.device("cuda", d)
        w = phant_amd.witness.block_witness(scale=args.block_scale, seed=4, device=dev, rank=d, world=D,
                                            ctx=phant_amd.Context(d), share=share if D > 1 else None)
        torch.cuda.synchronize()
        return w, dev

    if D > 1:
        for d in range(D):
            build(d, lambda c: (contribs.append(c.cpu().clone()), c)[1])
        level1 = sum(contribs[1:], contribs[0].clone())
    for d in range(D):
        w, dev = build(d, lambda c: level1.to(c.device))
        slots[d] = (w, torch.empty(w.batch.n, dtype=torch.uint8, device=dev),
                    torch.zeros(w.batch.n_roots, dtype=torch.int32, device=dev), comm.ctx(d))
    torch.cuda.set_device(0)
    n_roots = slots[0][0].batch.n_roots
    total = sum(s[0].batch.n for s in slots)

    def one_pass():
        for w, status, fails, c in slots:
            M.verify_batch_dev(w.batch, status=status, ctx=c, fail_count=fails)
        comm.allreduce_verdict([s[2] for s in slots], n_roots)

    def sync():
        for _, _, _, c in slots:
            c.sync()

    for _ in range(max(1, warmup) * inner):
        one_pass()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps * inner):
        one_pass()
    sync()
    elapsed = time.perf_counter() - t0
    want = sum(s[0].n_invalid for s in slots)
    for w, status, fails, _ in slots:
        assert torch.equal(status, w.expected), "verify statuses differ from the constructed expectation"
        assert int(fails.sum().item()) == want, (int(fails.sum().item()), want)
    passes = steps * inner
    line = {"metric": "mpt_proofs_verified_per_sec_block_witness", "value": total * passes / elapsed, "unit": "proofs/s",
            "n_gpus": D, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "ms_per_pass": elapsed / passes * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"config4 through phant_comm: one synthetic {int(10000 * args.block_scale)}-tx block witness "
                                   f"({total} account + storage proofs, {n_roots} roots) split over {D} device(s) of ONE process, "
                                   f"shards resident, one launch sequence per device and one phant_comm_allreduce_verdict per pass, "
                                   f"{inner} back-to-back passes per timed step",
                       "units_per_step": total, "parallelism": f"phant_comm x{D} (single process)",
                       "passes_per_timed_step": inner, "timed_region_ms": elapsed * 1e3},
            "devices": [bytes(device_id_bytes(torch.device("cuda", d)).cpu().tolist()).hex() for d in comm.devices],
            "expected": STRONG_EXPECTED}
    print(json.dumps(line), flush=True)
    comm.close()


def per_kernel_roofline(kernel_ms, tiers, n, w, valu_peak, form):
    """roofline.kernels: every stage of the verify launch against ITS roofline, from the serialised launch's events."""
    S = tiers["dedup_levels"]
    shallow = n * min(S, w.nodes_per_proof)  # nodes of the shallow tier
    copies = shallow - tiers["list_nodes"]
    node_b, key_b = 532, 32
    out = {"note": "tiers serialised (phant_diag_set: verify_serial): HIP events around each stage, alone on the chip; in one launch the "
                   "deep tier runs next to the shallow tier's propose -> dedup -> hash_list chain", "dedup_levels": S, "form": form}

    def hbm(ms, nbytes, what):
        g = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"ms": ms, "bound": "hbm", "algorithmic_bytes": int(nbytes), "achieved": g, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": g / HBM_PEAK_GBS, "bytes": what}

    def valu(ms, perms):
        g = perms / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"ms": ms, "bound": "valu", "keccak_f": int(perms), "achieved": g, "peak": valu_peak, "unit": "G Keccak-f/s",
                "frac": g / valu_peak}

    out["hash_deep_kernel"] = valu(kernel_ms["hash_deep"], tiers["deep_keccak_f"])
    out["hash_list_kernel"] = valu(kernel_ms["hash_list"], tiers["list_keccak_f"])
    out["dedup_kernel"] = hbm(kernel_ms["dedup"], copies * node_b + shallow * 16 + n * (key_b + 8),
                              "the copies' own bytes once (their representatives are cache hits) + offsets + keys")
    out["propose_kernel"] = hbm(kernel_ms["propose"], shallow * 16 + n * (key_b + 8) + tiers["list_nodes"] * 4,
                                "offsets + keys + table stores")
    out["walk_kernel"] = hbm(kernel_ms["walk"], n * (112 + key_b + 8 + 9) + w.nodes_per_proof * n * (1 + 4 * S // 8),
                             "the leaf, the key, node states and representatives, one status byte")
    return out


def mk_ctx(args, local_rank, use_torch_stream=True):
    import phant_amd
    c = phant_amd.Context(local_rank, use_torch_stream=use_torch_stream, verify_nodedup=(args.verify_mode == "nodedup"),
                          dedup_levels=(args.dedup_levels if args.verify_mode == "flat" else None))
    for item in filter(None, (x.strip() for x in args.diag.split(","))):
        k, _, v = item.partition("=")
        c.diag_set(k.strip(), int(v or "1"))
    return c


_SLOT_STREAMS = []


def run_proof_bench(args, make_witness, S, steps, warmup, inner, dev, local_rank, world, rank, ctx):
    """A resident proof batch verified + reduced to the per-root verdict, S launch sequences in flight, slot k on its
    own HIP stream / ctx (workspace) / status + verdict buffers and over its OWN witness (seed base + k: no pass
    re-reads what the previous pass on its slot left in L2 / Infinity Cache).  -> dict of measurements."""
    import torch
    import torch.distributed as dist
    from phant_amd import mpt as M

    n_wit = max(S, 2)
    wits = [make_witness(k) for k in range(n_wit)]
    K = max(1, min(args.allreduce_every, inner))  # passes per verdict exchange (1: a verdict per witness, what a validator needs)
    w0 = wits[0]
    n_units = w0.batch.n
    for w in wits:
        assert w.batch.n == n_units and w.batch.n_roots == w0.batch.n_roots
    slots = []
    torch.cuda.synchronize()
    for k in range(S):
        if k == 0:
            st_, c_ = torch.cuda.current_stream(dev), ctx
        else:
            # (the slot streams are made once per process: a second set -- the `strong` leg after the headline one -- would
            # be other entries of torch's stream pool, and slots whose streams share a hardware queue do not overlap)
            while len(_SLOT_STREAMS) < k:
                _SLOT_STREAMS.append(torch.cuda.Stream(device=dev))
            st_ = _SLOT_STREAMS[k - 1]
            with torch.cuda.stream(st_):
                c_ = mk_ctx(args, local_rank)
        # the slot's verdicts: two blocks of K passes x n_roots counters -- pass i of a block writes row i, a full block is
        # exchanged as ONE all-reduce while the passes of the other block go on
        slots.append((st_, c_, torch.empty(n_units, dtype=torch.uint8, device=dev),
                      torch.zeros((2, K, w0.batch.n_roots), dtype=torch.int32, device=dev), wits[k]))
    turn = {"k": 0}
    # The exchange (RCCL) on a stream of its own, fed by events: a collective issued on a slot's stream would put RCCL's
    # kernels and stream waits between that slot's launches -- and the hardware queues are a scarce resource here (DESIGN.md
    # section 7.5: a ninth stream / a second set of slot streams cost 10-20 %).
    comm_stream = torch.cuda.Stream(device=dev) if world > 1 else None
    wrote = [[[False] * K for _ in range(2)] for _ in range(S)]   # rows of the verdict blocks a pass has written
    filled = [0] * S                       # passes written into the slot's current block
    block_of = [0] * S                     # which block that is
    reduced_ev = [[None, None] for _ in range(S)]   # event behind the last exchange of block b of slot k
    exchanged = {"n": 0}

    def exchange(k):
        st_, _, _, fails_, _ = slots[k]
        b = block_of[k]
        if world > 1 and filled[k]:
            ready = torch.cuda.Event()
            ready.record(st_)
            comm_stream.wait_event(ready)
            with torch.cuda.stream(comm_stream):
                dist.all_reduce(fails_[b, :filled[k]])  # pass/fail words per root, over xGMI (RCCL)
                done = torch.cuda.Event()
                done.record(comm_stream)
            reduced_ev[k][b] = done
            exchanged["n"] += 1
        block_of[k], filled[k] = b ^ 1, 0
        if reduced_ev[k][b ^ 1] is not None:   # the block about to be overwritten: its exchange must be through
            st_.wait_event(reduced_ev[k][b ^ 1])
            reduced_ev[k][b ^ 1] = None

    def one_pass():
        k = turn["k"] % S
        st_, c_, status_, fails_, w_ = slots[k]
        turn["k"] += 1
        with torch.cuda.stream(st_):
            # statuses + per-root verdict of this pass
            M.verify_batch_dev(w_.batch, status=status_, ctx=c_, fail_count=fails_[block_of[k], filled[k]])
        wrote[k][block_of[k]][filled[k]] = True
        filled[k] += 1
        if filled[k] == K:
            exchange(k)

    def flush():
        for k in range(S):
            if filled[k]:
                exchange(k)
        if comm_stream is not None:
            comm_stream.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    # setup, not warm-up: every slot's ctx allocates its device workspace on its first call (hipMalloc synchronises
    # the device); keep that out of the W warm-up steps and the K timed steps
    for _ in range(S):
        one_pass()
    flush()
    torch.cuda.synchronize()
    for _ in range(warmup):
        one_pass()
    flush()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps * inner):
        one_pass()
    flush()  # (the last, possibly short, blocks: every verdict of the timed passes has been exchanged when the clock stops)
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    # correctness of what was timed: the statuses, and every row of the verdict blocks = the witness's failures over ALL ranks
    for k, (_, _, status_, fails_, w_) in enumerate(slots):
        assert torch.equal(status_, w_.expected), "verify statuses differ from the constructed expectation"
        want = torch.tensor([w_.n_invalid], dtype=torch.int32, device=dev)
        if world > 1:
            dist.all_reduce(want)
        # every row a pass wrote holds exactly the witness's failures (zero included); nothing else was touched
        mask = torch.tensor(wrote[k], dtype=torch.bool, device=dev).reshape(-1)
        rows = fails_.sum(dim=2).reshape(-1)
        assert bool(mask.any()), "no verdict row was written"
        assert bool((rows[mask] == int(want.item())).all()), (rows[mask][:8].tolist(), int(want.item()))
        assert bool((rows[~mask] == 0).all()), "a verdict row nobody wrote is not zero"

    timed_status0 = slots[0][2].cpu().numpy().copy()  # slot 0's last timed pass over wits[0] (the buffer is reused below)

    # the same number of passes strictly one after the other on ONE stream, alternating between two witnesses
    barrier()
    st0, c0, status0, fails0, _ = slots[0]
    t1 = time.perf_counter()
    with torch.cuda.stream(st0):
        for k in range(steps * inner):
            M.verify_batch_dev(wits[k % n_wit].batch, status=status0, ctx=c0, fail_count=fails0[0, k % K])
            if world > 1 and (k % K == K - 1 or k == steps * inner - 1):
                dist.all_reduce(fails0[0, :k % K + 1])
    barrier()
    e1 = max_over_ranks(time.perf_counter() - t1)

    # device time of ONE launch of the path (all kernels of the pipeline, first start to last end): HIP events on the
    # launch stream around a run of launches issued back to back (alternating witnesses, no verdict exchange), divided
    # by their number.  (Per-launch event pairs with a host synchronisation after each -- phant_timing, kept as
    # `kernel_synced_avg_ms` -- let the device idle between launches and read 1-4 % longer, box to box.)
    dstatus = torch.empty(n_units, dtype=torch.uint8, device=dev)
    n_timed = max(10, min(steps * inner, 120))
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st0):
        for k in range(4):
            M.verify_batch_dev(wits[k % n_wit].batch, status=dstatus, ctx=ctx)
        ev0.record(st0)
        for k in range(n_timed):
            M.verify_batch_dev(wits[k % n_wit].batch, status=dstatus, ctx=ctx)
        ev1.record(st0)
    torch.cuda.synchronize()
    k_evt_ms = ev0.elapsed_time(ev1) / n_timed
    ctx.timing(True)
    kms = []
    for k in range(max(10, min(steps * inner, 60))):
        M.verify_batch_dev(wits[k % n_wit].batch, status=dstatus, ctx=ctx)
        kms.append(ctx.last_kernel_ms())
    ctx.timing(False)
    hashed = ctx.verify_stats()
    # every kernel of the pipeline alone on the chip: a second ctx with the tiers serialised (phant_diag_set: verify_serial), HIP
    # events around each kernel (phant_verify_kernel_ms), averaged over a few launches
    kernels = tiers = bound = None
    if args.verify_mode == "flat":
        tiers = ctx.verify_tier_stats()
        if tiers["dedup_levels"]:
            with torch.cuda.stream(st0):
                cs = mk_ctx(args, local_rank)
            cs.diag_set("verify_serial", 1)
            acc, reps = {}, 8
            with torch.cuda.stream(st0):
                for k in range(reps + 2):
                    M.verify_batch_dev(wits[0].batch, status=dstatus, ctx=cs)
                    if k >= 2:
                        for name, ms in cs.verify_kernel_ms().items():
                            acc[name] = acc.get(name, 0.0) + ms / reps
            kernels = acc
            kernels["form"] = cs.verify_form()
            cs.close()
            if args.workload == "config3" and args.proofs >= 50_000:
                try:
                    with torch.cuda.stream(st0):
                        bound = slots[0][1].verify_bound_experiment(wits[0].batch, 20)
                except Exception as exc:  # diagnostics never cost the headline its line
                    bound = {"error": repr(exc)}
    passes = steps * inner
    out = {"wits": wits, "status0": timed_status0, "kernels": kernels, "bound": bound if isinstance(bound, dict) and "together_ms" in bound else None, "tiers": tiers, "n_units": n_units, "elapsed": elapsed, "passes": passes, "ms_per_pass": elapsed / passes * 1e3,
           "value": n_units * world * passes / elapsed,
           "single": {"value": n_units * world * passes / e1, "ms_per_pass": e1 / passes * 1e3, "ms_per_step": e1 / steps * 1e3},
           "k_avg_ms": k_evt_ms, "k_synced_ms": sum(kms) / len(kms), "k_min_ms": min(kms), "hashed": hashed,
           "verdict_exchange": {"passes_per_allreduce": K, "allreduces_on_this_rank": exchanged["n"],
                                "stream": "own (event-fed)" if world > 1 else None}}
    for k, (st_, c_, _, _, _) in enumerate(slots):
        if c_ is not ctx:
            c_.close()
    return out


def relaunch_under_torchrun(args):
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.environ.get("PHANT_BENCH_ENTRY", os.path.abspath(__file__)), *sys.argv[1:]]
    # (PHANT_BENCH_ENTRY: the CPU test suite's entry script -- this file's main() with torch.cuda stubbed out and gloo for RCCL)
    sys.stderr.write("bench.py: --gpus %d without WORLD_SIZE: re-executing as `%s`\n" % (args.gpus, " ".join(cmd)))
    sys.stderr.flush()
    try:
        rc = subprocess.run(cmd, timeout=(args.max_seconds + 120) if args.max_seconds > 0 else None).returncode
    except subprocess.TimeoutExpired:
        print(json.dumps({"metric": "mpt_proofs_verified_per_sec_depth%d" % args.depth, "value": None, "unit": "proofs/s",
                          "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                          "error": "the torch.distributed.run launch did not finish in time"}), flush=True)
        rc = 4
    sys.exit(rc)


def extra_legs(args):
    """The default run's short extra legs, each a bench.py process of its own (own ctx, own JSON line with its own `roofline`
    and oracle check), so that the driver's record of the default command carries more than config 3."""
    import subprocess
    legs = [("config2", ["--workload", "config2", "--steps", "10", "--warmup", "2", "--cpu-seconds", "3"]),
            ("mptize_1M_keys", ["--workload", "mptize", "--keys", "1000000", "--steps", "10", "--warmup", "2", "--cpu-seconds", "3"]),
            ("block_roots_100_items", ["--workload", "block_roots", "--items", "100", "--steps", "20", "--warmup", "3", "--cpu-seconds", "2"]),
            # (three launch sequences in flight, each over its own node set: 572 / 602 / 644 / 642 M keys/s at 1 / 2 / 3 / 4)
            ("nodeset_config3", ["--workload", "nodeset", "--streams", "3", "--steps", "10", "--warmup", "2", "--cpu-seconds", "3"]),
            # BASELINE config 5 at its stated length: 256 consecutive block witnesses (4 distinct ones in rotation), and the same
            # witnesses as node sets -- the form an execution witness has
            ("config5_256_blocks", ["--workload", "config5", "--steps", "64", "--warmup", "2", "--cpu-seconds", "3"]),
            ("config5_nodeset_256_blocks", ["--workload", "config5", "--nodeset", "--steps", "64", "--warmup", "2", "--cpu-seconds", "3"])]
    out, t_end = {}, time.time() + args.extra_seconds
    for name, argv in legs:
        left = t_end - time.time()
        if left < 30:
            out[name] = {"error": "skipped: the extra legs' wall-clock budget is spent"}
            continue
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extra", "--max-seconds", str(int(left)), *argv]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=left + 30)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            out[name] = json.loads(lines[-1]) if lines else {"error": f"no JSON line (rc {r.returncode})", "stderr_tail": r.stderr[-400:]}
        except Exception as e:  # a leg must never cost the headline its line
            out[name] = {"error": f"{type(e).__name__}: {e}"}
    return out


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    def give_up(msg, rc):
        # never a hang, never a silent death: the driver gets a line it can parse and a non-zero exit code
        if rank == 0:  # (straight to fd 1: nothing buffered is lost by the _exit below)
            os.write(1, (json.dumps({"metric": "mpt_proofs_verified_per_sec_depth%d" % args.depth, "value": None, "unit": "proofs/s",
                                     "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "error": msg}) + "\n").encode())
        sys.stderr.write(f"bench.py rank {rank}: {msg}\n")
        sys.stderr.flush()
        os._exit(rc)

    if args.max_seconds > 0:
        # a thread, not SIGALRM: a Python signal handler does not run while the main thread sits inside a blocking C call
        # (a collective that never completes, a stream synchronisation on a hung device)
        import threading
        guard = threading.Timer(args.max_seconds, lambda: give_up(f"wall-clock guard: not finished after {args.max_seconds} s", 4))
        guard.daemon = True
        guard.start()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and not args.comm:
        # started the way the N = 1 run is (`python bench.py --gpus N`): become the launcher -- one rank per GPU under
        # torch.distributed.run, same arguments; its rank 0 prints the line, this process hands on the exit code
        return relaunch_under_torchrun(args)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if args.comm:
        if world != 1:
            raise SystemExit("--comm is the single-process form: no torchrun")
        return run_comm_bench(args)
    rccl_world = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            # (RCCL; the CPU test suite drives the same code over gloo with torch.cuda stubbed out: tests/test_bench_emulated.py)
            backend = os.environ.get("PHANT_BENCH_BACKEND", "nccl")
            dist.init_process_group(backend, timeout=datetime.timedelta(seconds=args.rendezvous_seconds),
                                    **({"device_id": dev} if backend == "nccl" else {}))
            if backend == "gloo" and getattr(dev, "type", "cpu") == "cuda":
                # gloo on a GPU box (--one-device): device tensors go through host memory, in place
                raw_reduce, raw_gather = dist.all_reduce, dist.all_gather

                def all_reduce_staged(t, *a, **k):
                    if not t.is_cuda:
                        return raw_reduce(t, *a, **k)
                    h = t.cpu()
                    raw_reduce(h, *a, **k)
                    t.copy_(h)

                def all_gather_staged(out, t, *a, **k):
                    if not t.is_cuda:
                        return raw_gather(out, t, *a, **k)
                    hs = [torch.empty_like(t, device="cpu") for _ in out]
                    raw_gather(hs, t.cpu(), *a, **k)
                    for o, h in zip(out, hs):
                        o.copy_(h)

                dist.all_reduce, dist.all_gather = all_reduce_staged, all_gather_staged
            # who is in the job: every rank's device, gathered over the same backend the verdicts will take -- N distinct
            # UUIDs = N GPUs really took part (the first collective: RCCL builds its communicator here)
            mine = device_id_bytes(dev)
            ids = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(ids, mine)
            torch.cuda.synchronize()
        except Exception as e:  # rendezvous timed out, RCCL could not build the communicator, ...
            give_up(f"process group / RCCL initialisation failed: {type(e).__name__}: {e}", 3)
        uu = [bytes(t.cpu().tolist()).hex() for t in ids]
        rccl_world = {"ranks": world, "distinct_devices": len(set(uu)), "device_ids": uu, "backend": dist.get_backend()}

    import phant_amd
    from phant_amd import mpt as M
    from phant_amd.crypto import hasher as H

    ctx = mk_ctx(args, local_rank)  # bound to torch's current stream on this device

    proofs_like = args.workload in ("config3", "config4")  # a resident proof batch, verified + per-root verdict
    streamed = args.workload == "config5"
    single = None
    extra = {}
    strong = None
    S = 1
    inner = 1
    if proofs_like:
        S = max(1, min(args.streams, 8))
        inner = max(1, args.inner)

        def mk3(k):
            return phant_amd.witness.account_witness(args.proofs, depth=args.depth, seed=2 + k, device=dev, rank=rank,
                                                     world=world, ctx=ctx, key_order=args.proof_order)

        def mk4(k):
            return phant_amd.witness.block_witness(scale=args.block_scale, seed=4 + k, device=dev, rank=rank, world=world,
                                                   ctx=ctx)

        r = run_proof_bench(args, mk3 if args.workload == "config3" else mk4, S, args.steps, args.warmup, inner, dev,
                            local_rank, world, rank, ctx)
        w = r["wits"][0]
        b = w.batch
        n_units, elapsed, value, single, k_avg_ms = r["n_units"], r["elapsed"], r["value"], r["single"], r["k_avg_ms"]
        alg_bytes = b.algorithmic_bytes()
        ms_per_pass = r["ms_per_pass"]
        ms_per_step = ms_per_pass * inner
        timed_status0 = r["status0"]
        if args.workload == "config3":
            metric, unit = "mpt_proofs_verified_per_sec_depth%d" % args.depth, "proofs/s"
            workload = (f"config3: {args.proofs} synthetic depth-{args.depth} account proofs per GPU against one "
                        f"state root ({w.nodes_per_proof - 1} x 532 B full branches + 112 B leaf, "
                        f"{w.bytes_per_proof} B and {w.perms_per_proof} Keccak-f per proof, 1% corrupted/exclusion, "
                        f"no cross-proof dedup); every launch sequence in flight verifies its own witness (seeds 2.."
                        f"{1 + max(S, 2)}: distinct data per slot), {inner} back-to-back passes per timed step"
                        + ("" if args.proof_order == "random" else "; PROOFS IN ASCENDING KEY ORDER (not BASELINE's order)"))
        else:
            metric, unit = "mpt_proofs_verified_per_sec_block_witness", "proofs/s"
            workload = (f"config4: one synthetic {int(10000 * args.block_scale)}-tx block witness sharded over "
                        f"{world} GPU(s): {n_units} account + storage proofs on this rank against {b.n_roots} roots "
                        f"(state root + per-contract storage roots; depth 8 / 3 / 5 / 7 classes, "
                        f"{w.nodes_per_proof:.2f} nodes, {w.bytes_per_proof:.0f} B and {w.perms_per_proof:.1f} "
                        f"Keccak-f per proof on average, 1% corrupted/exclusion); distinct witness per slot, "
                        f"{inner} back-to-back passes per timed step")
        extra = {"kernel_synced_avg_ms": r["k_synced_ms"], "verdict_exchange": r["verdict_exchange"]}
        if r["hashed"] is not None:
            hashed = r["hashed"]
            kf = int(sum((c + 1) * h for c, h in enumerate(hashed)))
            # the second roofline of this path: Keccak-f is integer-VALU-bound.  Peak = what the product's round function
            # sustains with nothing but permutations on THIS chip, measured in this run (phant_keccak_rate: 6 waves per SIMD,
            # ~5 ms; 4 = what the 120-VGPR hash kernels can have).  Round 1's stand-alone figure: 10.3 G perm/s
            # (profiles/r1i/keccak_rate_ubench.txt)
            peak6 = ctx.keccak_rate(6, 200) / 1e9
            peak4 = ctx.keccak_rate(4, 200) / 1e9
            vpeak = max(peak6, peak4)
            extra = {**extra, "nodes_shipped": int(b.node_off.numel() - 1), "nodes_hashed": int(sum(hashed)), "keccak_f_run": kf,
                     "keccak_f_if_every_node_hashed": int(round(w.perms_per_proof * n_units)),
                     "valu": {"bound": "valu", "achieved": kf / (k_avg_ms * 1e-3) / 1e9, "peak": vpeak,
                              "unit": "G Keccak-f/s", "frac": kf / (k_avg_ms * 1e-3) / 1e9 / vpeak,
                              "peak_source": "phant_keccak_rate on this device, this run: permutations only, the better of 4 and 6 "
                                             "waves per SIMD",
                              "peak_at_6_waves_per_simd": peak6, "peak_at_4_waves_per_simd": peak4, "peak_round1_ubench": 10.3,
                              "note": "permutations actually run / whole-pipeline time of one launch"}}
            if args.workload == "config3" and args.verify_mode == "flat" and r.get("kernels"):
                extra["kernels"] = per_kernel_roofline(r["kernels"], r["tiers"], n_units, w, vpeak, r["kernels"].get("form"))
            if args.workload == "config3" and args.verify_mode == "flat" and r.get("bound"):
                # What the chip overlaps at best on this witness (phant_verify_bound_experiment): the launch's hashing alone, a
                # clean coalesced read of its bytes alone, both next to each other -- launched and timed the way one launch is
                # (events on the launch stream around back-to-back repetitions, the same fork / join of helper streams).  A verify
                # launch cannot be shorter than together_ms: ceiling_frac is the HBM fraction of THAT time.
                be = dict(r["bound"])
                be["ceiling_frac"] = alg_bytes / (be["together_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                be["one_launch_over_ceiling"] = k_avg_ms / be["together_ms"]
                be["note"] = ("hash_only_ms / stream_only_ms / together_ms of the same witness; together_ms >= 0.19 ms: the 40 % bar "
                              "(0.121 ms) is out of reach for this hashing next to ANY stream on this chip -- what is left between one "
                              "launch and together_ms is all the shallow tier's own shape can still give")
                extra["bound_experiment"] = be
        if args.workload == "config3" and not args.no_strong:
            # BASELINE config 4 next to it: ONE block witness split over the same N GPUs (strong scaling): accounts by
            # top key nibble, contracts dealt out whole, one all-reduce of the per-root verdicts per pass
            del r
            torch.cuda.synchronize()
            r4 = run_proof_bench(args, mk4, S, max(2, args.steps // 2), 1, inner, dev, local_rank, world, rank, ctx)
            w4 = r4["wits"][0]
            tot = torch.tensor([r4["n_units"]], dtype=torch.int64, device=dev)
            if world > 1:
                dist.all_reduce(tot)
            total_proofs = int(tot.item())
            strong = {"workload": f"config4: one synthetic {int(10000 * args.block_scale)}-tx block witness "
                                  f"({total_proofs} account + storage proofs, {w4.batch.n_roots} roots) split over "
                                  f"{world} GPU(s)", "scaling": "strong", "metric": "mpt_proofs_verified_per_sec_block_witness",
                      "value": total_proofs * r4["passes"] / r4["elapsed"], "unit": "proofs/s",
                      "ms_per_pass": r4["ms_per_pass"], "ms_per_step": r4["ms_per_pass"] * inner, "proofs_on_rank0": r4["n_units"],
                      "single_stream": {"value": total_proofs * r4["passes"] / (r4["single"]["ms_per_pass"] * 1e-3 * r4["passes"]),
                                        "ms_per_pass": r4["single"]["ms_per_pass"]},
                      "kernel_avg_ms": r4["k_avg_ms"],
                      "roofline_frac": w4.batch.algorithmic_bytes() / (r4["k_avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                      "nodes_hashed": int(sum(r4["hashed"])), "nodes_shipped": int(w4.batch.node_off.numel() - 1),
                      "verdict_exchange": r4["verdict_exchange"], "expected": STRONG_EXPECTED}
            strong["predicted"] = strong_predicted(total_proofs, world, strong["value"])
            del r4
    elif args.workload == "nodeset":
        # config 3's trie as a node SET: S launch sequences in flight (slot k: own ctx, torch stream, witness of seed 2 + k), every
        # pass = phant_mpt_verify_nodeset_verdict_dev (statuses + the per-root verdict out of one launch) + the verdict's all-reduce
        S = max(1, min(args.streams, 4))
        ns_slots = []
        for k in range(S):
            stream_k = torch.cuda.current_stream(dev) if k == 0 else torch.cuda.Stream(device=dev)
            with torch.cuda.stream(stream_k):
                c_k = ctx if k == 0 else mk_ctx(args, local_rank)
                w_k = phant_amd.witness.account_witness(args.proofs, depth=args.depth, seed=2 + k, device=dev, rank=rank, world=world,
                                                        ctx=c_k, corrupt_frac=0.01)
                s_k = phant_amd.witness.node_set(w_k, ctx=c_k, shuffle_seed=7 + k)
            ns_slots.append((stream_k, c_k, w_k, s_k, torch.empty(s_k.n, dtype=torch.uint8, device=dev),
                             torch.zeros(1, dtype=torch.int32, device=dev)))
        torch.cuda.synchronize()
        w, sset = ns_slots[0][2], ns_slots[0][3]
        b = w.batch
        n_units = sset.n
        alg_bytes = sset.algorithmic_bytes()
        ns_turn = {"k": 0}

        def kernel_only():
            M.verify_nodeset_dev(sset.roots, None, sset.keys, sset.nodes, sset.node_off, status=ns_slots[0][4], ctx=ctx)

        def step():
            st_, c_, _, s_, status_, fails_ = ns_slots[ns_turn["k"] % S]
            ns_turn["k"] += 1
            with torch.cuda.stream(st_):
                M.verify_nodeset_dev(s_.roots, None, s_.keys, s_.nodes, s_.node_off, status=status_, ctx=c_, fail_count=fails_)
                if world > 1:
                    dist.all_reduce(fails_)

        metric, unit = "mpt_keys_verified_per_sec_depth%d_nodeset" % args.depth, "proofs/s"
        workload = (f"nodeset: {args.proofs} depth-{args.depth} keys per GPU against one state root, witness = the "
                    f"{sset.total_nodes} distinct nodes shipped once, in random order ({sset.nodes.numel()} B; 1% of the source "
                    f"proofs damaged / exclusion proofs: a damaged copy is a node nobody refers to), {S} launch sequence(s) in "
                    f"flight, each over its own witness")
    elif args.workload == "mptize":
        # the state-trie hasher: n sorted distinct random 32-byte keys (hashed addresses), 78-byte values (account RLP)
        n_units = args.keys
        g = torch.Generator(device=dev)
        g.manual_seed(7 + rank)
        k64 = torch.randint(-(1 << 62), 1 << 62, (int(n_units * 1.01) + 16, 4), dtype=torch.int64, device=dev, generator=g)
        # sort by the big-endian byte order of the key = lexicographic order of the four words taken as unsigned
        kb = k64.view(torch.uint8).reshape(-1, 32)
        be = torch.flip(kb.reshape(-1, 4, 8), dims=[2]).contiguous().view(torch.int64).reshape(-1, 4)  # words as BE values
        order = torch.arange(be.shape[0], device=dev)
        for c in (3, 2, 1, 0):
            u = be[order, c]
            order = order[torch.argsort(u ^ (-(1 << 63)), stable=True)]  # unsigned order
        kb = kb[order]
        keep = torch.ones(kb.shape[0], dtype=torch.bool, device=dev)
        keep[1:] = (kb[1:] != kb[:-1]).any(dim=1)
        kb = kb[keep][:n_units].contiguous()
        assert kb.shape[0] == n_units
        keys_t = kb.reshape(-1)
        key_off = (torch.arange(n_units + 1, device=dev, dtype=torch.int64) * 32).to(torch.int32)
        vals_t = torch.randint(0, 256, (n_units * 78,), dtype=torch.uint8, device=dev, generator=g)
        val_off = torch.arange(n_units + 1, device=dev, dtype=torch.int64) * 78
        root = torch.empty(32, dtype=torch.uint8, device=dev)
        alg_bytes = int(keys_t.numel() + vals_t.numel() + 32)
        torch.cuda.synchronize()

        def step():
            M.mptize_dev(keys_t, key_off, vals_t, val_off, out=root, ctx=ctx)

        kernel_only = step
        trie_perms, trie_nodes = trie_keccak_f(kb, 78)
        metric, unit = "mpt_trie_keys_hashed_per_sec", "keys/s"
        workload = (f"mptize: root of the trie of {n_units} sorted random 32-byte keys with 78-byte values per GPU, arrays "
                    f"resident in HBM (phant_mpt_root_dev; {alg_bytes} B = keys + values + root)")
    elif args.workload == "block_roots":
        # what the reference's live callers of mptize compute per block (src/blockchain/blockchain.zig:198-204,209-235): the
        # index-keyed roots of its transaction / receipt / withdrawal lists -- here as ONE phant_block_roots call, host form (the
        # caller's bytes in, three roots out: copies included -- at this size the call is launch latency, not bandwidth)
        import numpy as np
        rngb = np.random.default_rng(3 + rank)
        mkl = lambda lo, hi: [rngb.integers(0, 256, int(rngb.integers(lo, hi)), dtype=np.uint8).tobytes() for _ in range(args.items)]  # noqa: E731
        lists = [mkl(100, 300), mkl(300, 700), mkl(40, 60)]
        n_units = 3
        alg_bytes = int(sum(len(x) for l_ in lists for x in l_) + 3 * 32)
        roots_got = {}
        # (packed once: a step is the C call, not Python's marshalling of 300 byte strings)
        import ctypes as C
        from phant_amd.mpt import _pack, _np_ptr
        packed = [_pack(x, np.uint64) for x in lists]
        item_p = (C.c_void_p * 3)(*[_np_ptr(b_).value for b_, _ in packed])
        off_p = (C.c_void_p * 3)(*[_np_ptr(o_).value for _, o_ in packed])
        cnt_p = (C.c_uint32 * 3)(*[len(x) for x in lists])
        roots_out = np.zeros(96, np.uint8)

        def step():
            ctx.check(ctx._lib.phant_block_roots(ctx.handle, item_p, off_p, cnt_p, 3, _np_ptr(roots_out), None, None, None, 0, 0, None))
            roots_got["r"] = [roots_out[32 * i:32 * i + 32].tobytes() for i in range(3)]

        kernel_only = None
        metric, unit = "mpt_block_index_roots_per_sec", "roots/s"
        workload = (f"block_roots: the three index-keyed roots of a block (transactions / receipts / withdrawals, {args.items} items "
                    f"each, {alg_bytes} B) in one phant_block_roots call, host form (copies in and out included)")
    elif args.workload == "config5":
        # 4 distinct witnesses in pinned host memory, submitted round-robin; results land in pinned buffers
        from phant_amd import mpt as MM
        if args.stream_proofs:
            wl = [phant_amd.witness.account_witness(args.stream_proofs, depth=args.depth, seed=40 + k, device=dev,
                                                    rank=rank, world=world, ctx=ctx) for k in range(4)]
        else:  # BASELINE config 5: block witnesses (config 4's shape; this rank's share of each)
            wl = [phant_amd.witness.block_witness(scale=args.block_scale, seed=40 + k, device=dev, rank=rank, world=world,
                                                  ctx=ctx) for k in range(4)]
        if args.nodeset:  # the same witnesses as node SETS: every distinct node crosses the bus once
            sets = [phant_amd.witness.node_set(x, ctx=ctx, shuffle_seed=70 + k) for k, x in enumerate(wl)]
            hosts = [MM.nodeset_to_host(x) for x in sets]
            sset = sets[0]
        else:
            hosts = [MM.to_host(x.batch) for x in wl]
        w = wl[0]
        b = w.batch
        n_units = b.n
        for x in wl:
            assert x.batch.n == n_units
        alg_bytes = sets[0].algorithmic_bytes() if args.nodeset else wl[0].batch.algorithmic_bytes()
        slots = max(1, min(args.stream_slots, 4))
        state = {"k": 0, "pending": []}

        def step():
            k = state["k"]
            if len(state["pending"]) == slots:
                MM.wait(state["pending"].pop(0), ctx)
            if args.nodeset:
                MM.verify_nodeset_submit(hosts[k % 4], k % slots, ctx)
            else:
                MM.verify_submit(hosts[k % 4], k % slots, ctx)
            state["pending"].append(k % slots)
            state["k"] = k + 1

        def drain():
            while state["pending"]:
                MM.wait(state["pending"].pop(0), ctx)

        kernel_only = None
        if args.stream_proofs:
            metric, unit = "mpt_proofs_verified_per_sec_depth%d_streamed" % args.depth, "proofs/s"
            what = f"witnesses of {args.stream_proofs} depth-{args.depth} account proofs"
        else:
            metric, unit = "mpt_proofs_verified_per_sec_block_witness_streamed", "proofs/s"
            what = (f"synthetic {int(10000 * args.block_scale)}-tx block witnesses ({n_units} account + storage proofs against "
                    f"{b.n_roots} roots on this rank each; 4 distinct ones in rotation)")
        workload = (f"config5: consecutive {what} streamed from pinned host memory, {slots} in flight per GPU "
                    f"(H2D {hosts[0].h2d_bytes()} B per witness)")
        if args.nodeset:
            metric += "_nodeset"
            workload += (f"; every witness shipped as a node SET ({sets[0].total_nodes} distinct nodes of the "
                         f"{int(b.node_off.numel() - 1)} the per-proof form ships, random order: the form an execution witness has) "
                         f"through phant_mpt_verify_nodeset_submit")
    else:
        n_units = args.messages
        g = torch.Generator(device=dev)
        g.manual_seed(1 + rank)
        blob = torch.randint(0, 256, (n_units * 136,), dtype=torch.uint8, device=dev, generator=g)
        out = torch.empty((n_units, 32), dtype=torch.uint8, device=dev)
        alg_bytes = n_units * 168

        def step():
            H.keccak256_fixed_dev(blob, 136, n_units, out=out, ctx=ctx)

        kernel_only = step
        metric, unit = "keccak256_136B_hashes_per_sec", "hashes/s"
        workload = f"config2: {n_units} x 136-byte Keccak-256 per GPU (2 Keccak-f per message, 168 B per message)"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if not proofs_like:
        # back-to-back repetitions per timed step (timed region of the order of 100 ms; figures are per single pass)
        inner = {"config2": max(1, args.inner), "nodeset": max(1, args.inner), "config5": 4, "mptize": 4, "block_roots": 10}[args.workload]
        for _ in range(args.warmup):
            step()
        if streamed:
            drain()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps * inner):
            step()
        if streamed:
            drain()
        barrier()
        elapsed = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        value = n_units * world * args.steps * inner / elapsed
        ms_per_pass = elapsed / (args.steps * inner) * 1e3
        ms_per_step = ms_per_pass * inner

        # correctness of what was timed
        cpu_check_status = None
        if args.workload == "nodeset":
            cpu_check_status = ns_slots[0][4].cpu().numpy().copy()  # slot 0's last timed pass over its witness
            for _, _, w_, _, status_, fails_ in ns_slots:
                assert torch.equal(status_, w_.expected_nodeset), "node-set statuses differ from the constructed expectation"
                want_f = torch.tensor([int((w_.expected_nodeset >= 16).sum().item())], dtype=torch.int32, device=dev)
                if world > 1:
                    dist.all_reduce(want_f)
                assert int(fails_.item()) == int(want_f.item()), (int(fails_.item()), int(want_f.item()))
        if streamed:
            for x, h in zip(wl, hosts):
                assert torch.equal(h.status, (x.expected_nodeset if args.nodeset else x.expected).cpu()), \
                    "streamed statuses differ from the expectation"
            # the kernels of one witness, device-resident, for the roofline object (the streamed rate itself is
            # bounded by PCIe: see the pcie object)
            dstatus = torch.empty(n_units, dtype=torch.uint8, device=dev)
            if args.nodeset:
                cpu_check_status = hosts[0].status.numpy().copy()

                def kernel_only():
                    M.verify_nodeset_dev(sset.roots, sset.root_idx, sset.keys, sset.nodes, sset.node_off, status=dstatus, ctx=ctx)
            else:
                def kernel_only():
                    M.verify_batch_dev(wl[0].batch, status=dstatus, ctx=ctx)

        # device time of one launch of the path (all kernels of the verify pipeline / the sponge kernel),
        # HIP events on the launch stream (phant_timing)
        if args.workload == "block_roots":
            # (a synchronous host-form call: its wall time IS the call; the trie's Keccak-f against the chip's rate says how far
            # from any throughput bound a block-sized trie is -- it is a chain of launch and sponge latencies)
            k_avg_ms = ms_per_pass
            from oracle import oracle as O
            assert roots_got["r"] == [O.index_root_rlp(x) for x in lists], "block roots differ from the oracle"
        else:
            ctx.timing(True)
            kms = []
            for _ in range(max(5, min(args.steps, 50))):
                kernel_only()
                kms.append(ctx.last_kernel_ms())
            ctx.timing(False)
            k_avg_ms = sum(kms) / len(kms)
        if args.workload == "nodeset":
            # one launch: HIP events on the launch stream around a run of back-to-back launches (alternating witnesses), as for
            # the per-proof line; the per-launch pairs above (a host synchronisation after each) stay as kernel_synced_avg_ms
            n_timed = max(10, min(args.steps * inner, 120))
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            st0 = ns_slots[0][0]
            sets = [x[3] for x in ns_slots]
            with torch.cuda.stream(st0):
                for k in range(4):
                    M.verify_nodeset_dev(sets[k % S].roots, None, sets[k % S].keys, sets[k % S].nodes, sets[k % S].node_off,
                                         status=ns_slots[0][4], ctx=ctx)
                ev0.record(st0)
                for k in range(n_timed):
                    M.verify_nodeset_dev(sets[k % S].roots, None, sets[k % S].keys, sets[k % S].nodes, sets[k % S].node_off,
                                         status=ns_slots[0][4], ctx=ctx)
                ev1.record(st0)
            torch.cuda.synchronize()
            synced = k_avg_ms
            k_avg_ms = ev0.elapsed_time(ev1) / n_timed
            kernel_only()  # (the statistics are the last launch's: slot 0's witness)
            hashed = ctx.verify_stats()
            kf = int(sum((c + 1) * h for c, h in enumerate(hashed)))
            peak6 = ctx.keccak_rate(6, 200) / 1e9
            peak4 = ctx.keccak_rate(4, 200) / 1e9
            vpeak = max(peak6, peak4)
            extra = {"kernel_synced_avg_ms": synced, "nodes_shipped": int(sset.total_nodes), "nodes_hashed": int(sum(hashed)),
                     "keccak_f_run": kf,
                     "valu": {"bound": "valu", "achieved": kf / (k_avg_ms * 1e-3) / 1e9, "peak": vpeak, "unit": "G Keccak-f/s",
                              "frac": kf / (k_avg_ms * 1e-3) / 1e9 / vpeak,
                              "throughput_frac": value / world / n_units * kf / 1e9 / vpeak,
                              "peak_source": "phant_keccak_rate on this device, this run: permutations only, the better of 4 and 6 "
                                             "waves per SIMD",
                              "note": "every node of the set is hashed exactly once: permutations run / whole-pipeline time of one "
                                      "launch (frac) and / the rate with the launch sequences in flight (throughput_frac)"}}
        if streamed:
            hashed = ctx.verify_stats()
            extra = {"nodes_shipped": int(sset.total_nodes if args.nodeset else b.node_off.numel() - 1), "nodes_hashed": int(sum(hashed))}
        if args.workload == "mptize":
            # the roofline that bounds a trie hasher: every node is hashed once, Keccak-f is integer-VALU-bound (110 MB of keys and
            # values are 1.6 % of HBM's rate at these times: the wrong ceiling).  Peak as for the verify line: measured in this run
            peak6 = ctx.keccak_rate(6, 200) / 1e9
            peak4 = ctx.keccak_rate(4, 200) / 1e9
            vpeak = max(peak6, peak4)
            extra = {"keccak_f_run": int(trie_perms), "trie_nodes": trie_nodes,
                     "valu": {"bound": "valu", "achieved": trie_perms / (k_avg_ms * 1e-3) / 1e9, "peak": vpeak, "unit": "G Keccak-f/s",
                              "frac": trie_perms / (k_avg_ms * 1e-3) / 1e9 / vpeak,
                              "peak_source": "phant_keccak_rate on this device, this run: permutations only, the better of 4 and 6 "
                                             "waves per SIMD",
                              "note": "one Keccak-f per node and rate block, counted from the trie's shape (bench.trie_keccak_f) / "
                                      "device time of one call (first kernel start to last kernel end)"}}
    achieved = alg_bytes / (k_avg_ms * 1e-3) / 1e9

    pipeline = ("two-tier verify pipeline = hash_deep_kernel (in-place hashing of the deep levels, helper stream) next to the "
                "shallow tier (propose_kernel + dedup_kernel + hash_list_kernel: the copies of the upper trie levels are byte-compared "
                "with one representative per group instead of hashed), then walk_kernel (one launch of the path, first kernel start "
                "to last kernel end; the hash kernels are integer-VALU-bound, see roofline.valu); a batch of less than 72 MB of nodes: "
                "zero_kernel + hash_deep / hash_coop / hash_wave_kernel over every node + walk_kernel")
    line = {
        "metric": metric, "value": value, "unit": unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": ms_per_step, "ms_per_pass": ms_per_pass, "higher_is_better": True,
        "scaling": "strong" if args.workload == "config4" else "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
        "config": {"workload": workload, "units_per_gpu_per_step": n_units, "parallelism": f"key-sharded x{world}",
                   **({"parity_basis": "derived: the reference has no verifier (TODO at src/engine_api/execution_payload.zig:177-178), "
                                       "so no known answer exists; statuses are checked against the constructed expectation here and, "
                                       "in tests/, against oracle/verify.c over proofs extracted from tries whose roots the "
                                       "reference's own vectors pin (DESIGN.md section 5)"} if proofs_like or streamed or
                      args.workload == "nodeset" else {}),
                   "verify_mode": args.verify_mode if proofs_like else None,
                   "dedup_levels": (args.dedup_levels if proofs_like else None),
                   "streams": (S if (proofs_like or args.workload == "nodeset") else 1), "passes_per_timed_step": inner,
                   "timed_region_ms": ms_per_step * args.steps,
                   "step": f"a timed step = {inner} back-to-back pass(es) of the path over a resident batch; ms_per_step x steps = the "
                           "timed region, ms_per_pass = one pass (`value` = units per pass / ms_per_pass)"},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS,
                     "throughput_GBps": value / world * alg_bytes / n_units / 1e9,  # algorithmic bytes x the measured
                     "throughput_frac": value / world * alg_bytes / n_units / 1e9 / HBM_PEAK_GBS,  # rate (launches overlap)
                     "traffic": (tr["bytes"] if (tr := (pmc_traffic(args.verify_mode, args.proofs) if args.workload == "config3" else
                                                        pmc_traffic("nodeset", args.proofs) if args.workload == "nodeset" else None)) else None),
                     "traffic_source": (tr.get("source") or tr.get("stale")) if tr else None,
                     "traffic_detail": tr,
                     "kernel": ("keccak256_fixed_kernel" if args.workload == "config2" else
                                "node-set pipeline = set_classify_kernel (class lists) + set_hash_kernel (every node hashed once and put "
                                "into the record table by the lane that hashed it, duplicates into a second table by the same lane) "
                                "+ set_walk_kernel (one launch of the path, first kernel start to last kernel end; the hash kernel is "
                                "integer-VALU-bound, see roofline.valu)" if args.workload == "nodeset" else
                                "trie hasher on three block-sized lists as one forest, the pass for small tries: small_head_kernel (one "
                                "workgroup: lcp, min-tree in LDS, the nodes) + small_climb_kernel (a wave per key: the leaf, then every node "
                                "whose last child the wave has just delivered); the call is as long as its longest chain of nodes + ~55 us of "
                                "launches, copies and a wake-up"
                                if args.workload == "block_roots" else
                                "trie hasher = head_kernel + lcp_kernel + tree_levels_kernel x 2 + identify_kernel + order_kernel + leaf_kernel (the keys under the deepest nodes first; the deepest depth bins beside the rest) + per depth bin branch_kernel<1|2|4> or, for a thin bin, branch_coop_kernel, finish_kernel (first start to last end)"
                                if args.workload == "mptize" else
                                "node-set pipeline = set_classify_kernel + set_hash_kernel + set_walk_kernel"
                                if (streamed and args.nodeset) else pipeline),
                     "kernel_avg_ms": k_avg_ms, "algorithmic_bytes_per_launch": alg_bytes, **extra},
    }
    # the scalars a reader of the compacted line needs next to `frac` ("report VALU utilisation next to the
    # HBM fraction"): the hashing is integer-VALU-bound, so the HBM fraction a launch could reach at most is the one it would have
    # if it took exactly as long as its permutations take at the chip's measured Keccak-f rate
    rf = line["roofline"]
    if isinstance(rf.get("valu"), dict) and rf["valu"].get("peak"):
        rf["valu_frac"] = rf["valu"]["frac"]
        t_valu_ms = rf["keccak_f_run"] / (rf["valu"]["peak"] * 1e9) * 1e3
        rf["valu_bound_frac_of_hbm"] = alg_bytes / (t_valu_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        rf["valu_bound_ms"] = t_valu_ms
    if isinstance(rf.get("bound_experiment"), dict) and "ceiling_frac" in rf["bound_experiment"]:
        rf["ceiling_frac"] = rf["bound_experiment"]["ceiling_frac"]
    if single is not None:
        line["single_stream"] = single
    if strong is not None:
        line["strong"] = strong
    if streamed:
        h2d = hosts[0].h2d_bytes() * world * args.steps * inner / elapsed / 1e9
        line["pcie"] = {"h2d_GBps_all_gpus": h2d, "h2d_GBps_per_gpu": h2d / world, "peak_per_gpu": 63.0,
                        "frac": h2d / world / 63.0, "slots": slots,
                        "note": "PCIe Gen5 x16 spec; the streamed rate is H2D-bound, the kernels of one witness "
                                "take roofline.kernel_avg_ms"}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        if args.workload == "mptize":
            line["cpu_baseline"] = cpu_baseline_mptize(keys_t, vals_t, n_units, root, args.cpu_seconds)
        elif args.workload == "block_roots":
            line["cpu_baseline"] = cpu_baseline_block_roots(lists, min(args.cpu_seconds, 3.0))
        elif args.workload == "nodeset" or (args.workload == "config5" and args.nodeset):
            line["cpu_baseline"] = cpu_baseline_nodeset(sset, args.cpu_seconds, cpu_check_status)
        elif args.workload == "config4":
            line["cpu_baseline"] = cpu_baseline_block(w, args.cpu_seconds)
        elif args.workload == "config5" and not args.stream_proofs:
            line["cpu_baseline"] = cpu_baseline_block(w, args.cpu_seconds)
        elif args.workload in ("config3", "config5"):
            line["cpu_baseline"] = cpu_baseline_config3(w, args.cpu_seconds, timed_status0 if args.workload == "config3" else None)
        else:
            line["cpu_baseline"] = cpu_baseline_config2(blob, n_units, args.cpu_seconds)
    if rccl_world is not None:
        line["rccl_world"] = rccl_world
    if (rank == 0 and world == 1 and args.workload == "config3" and not args.no_extra and args.proofs == 100_000 and
            args.verify_mode == "flat" and not args.no_strong and not args.no_cpu_baseline and args.proof_order == "random"):
        # (only the default-sized headline run: the sweeps and A/B scripts call with other sizes or --no-extra)
        line["extra"] = extra_legs(args)
        line["extra_keys"] = sorted(line["extra"])
    if rank == 0:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
