#!/usr/bin/env python3
"""Board power and shader clock (rocm-smi, ~4 samples per second) while phant_verify_bound_experiment runs its three phases long enough
to be seen: the launch's hashing alone, a clean read of the witness alone, both next to each other (BASELINE config 3).
Is the chip up against a power limit when the two run together?    python tools/probe_bound_power.py [reps]"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
w = phant_amd.witness.account_witness(100_000, depth=8, seed=2, device=dev, ctx=ctx)
print(subprocess.run(["rocm-smi", "--showmaxpower", "--showpower", "--json"], capture_output=True, text=True).stdout.strip()[:600], flush=True)
rows, stop = [], False


def sample():
    while not stop:
        t = time.time()
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10)
            d = json.loads(r.stdout)
            c = next(iter(d.values()))
            rows.append((t, c.get("Current Socket Graphics Package Power (W)") or c.get("Average Graphics Package Power (W)"), c.get("sclk clock speed:")))
        except Exception as e:  # noqa: BLE001
            rows.append((t, repr(e)[:60], None))
        time.sleep(0.1)


th = threading.Thread(target=sample, daemon=True)
th.start()
time.sleep(1.0)
t0 = time.time()
res = ctx.verify_bound_experiment(w.batch, reps)
t1 = time.time()
time.sleep(0.5)
stop = True
th.join(timeout=5)
print(json.dumps({"reps": reps, **res, "wall_s": t1 - t0}))
# the three phases by their share of the wall time (warm-up launches aside): hash | stream | together
tot = res["hash_only_ms"] + res["stream_only_ms"] + res["together_ms"]
b1 = t0 + (t1 - t0) * res["hash_only_ms"] / tot
b2 = t0 + (t1 - t0) * (res["hash_only_ms"] + res["stream_only_ms"]) / tot
for t, p, c in rows:
    ph = "idle" if t < t0 or t > t1 else "hash" if t < b1 else "stream" if t < b2 else "together"
    print(f"{t - t0:7.2f} s  {ph:9s} {p} W  sclk {c}")
