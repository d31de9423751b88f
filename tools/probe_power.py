#!/usr/bin/env python3
"""Shader clock and board power while the device runs (a) nothing, (b) nothing but Keccak-f, (c) a read-only memory stream,
(d) the verify pipeline (four launches in flight), (e) the same with every node hashed -- a few seconds each, sampled from
sysfs (hwmon freq1_input / power1_average, pp_dpm_sclk) and, if that is not there, from one `rocm-smi` call per phase.

    python tools/probe_power.py [--seconds 3] [--out gpurun_out/power.jsonl]

Why: three rounds of scheduling experiments fit an ADDITIVE model -- a launch takes its Keccak-f at the VALU ceiling plus its
bytes at ~10 TB/s, however the two are interleaved.  A board that runs into its power limit behaves
exactly like that (time = energy / cap); this says whether it does.
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def sysfs_sources():
    src = {}
    for card in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
        if not os.path.exists(os.path.join(card, "pp_dpm_sclk")):
            continue
        src["pp_dpm_sclk"] = os.path.join(card, "pp_dpm_sclk")
        src["pp_dpm_mclk"] = os.path.join(card, "pp_dpm_mclk")
        for h in glob.glob(os.path.join(card, "hwmon", "hwmon*")):
            for f in ("freq1_input", "freq2_input", "power1_average", "power1_input", "temp1_input", "temp2_input"):
                p = os.path.join(h, f)
                if os.path.exists(p):
                    src[f] = p
        break
    return src


def read_dpm(path):
    """the starred level of pp_dpm_*: MHz"""
    try:
        for line in open(path).read().splitlines():
            if line.rstrip().endswith("*"):
                return float(line.split(":")[1].lower().replace("mhz", "").replace("*", "").strip())
    except Exception:
        return None
    return None


def read_num(path):
    try:
        return float(open(path).read().strip())
    except Exception:
        return None


class Sampler(threading.Thread):
    def __init__(self, src, period=0.02):
        super().__init__(daemon=True)
        self.src, self.period, self.rows, self.stop_flag = src, period, [], False

    def run(self):
        while not self.stop_flag:
            row = {}
            for k, p in self.src.items():
                row[k] = read_dpm(p) if k.startswith("pp_dpm") else read_num(p)
            self.rows.append(row)
            time.sleep(self.period)

    def summary(self):
        out = {"samples": len(self.rows)}
        for k in self.src:
            v = [r[k] for r in self.rows if r.get(k) is not None]
            if v:
                scale = 1e-6 if k.startswith(("freq", "power")) else (1e-3 if k.startswith("temp") else 1.0)
                out[k] = {"mean": round(sum(v) / len(v) * scale, 1), "min": round(min(v) * scale, 1), "max": round(max(v) * scale, 1)}
        return out


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20)
        return json.loads(r.stdout) if r.stdout.strip().startswith("{") else r.stdout[-400:]
    except Exception as e:  # noqa: BLE001
        return repr(e)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=3.0)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import torch
    import phant_amd
    from phant_amd import mpt as M

    dev = torch.device("cuda", 0)
    src = sysfs_sources()
    print(json.dumps({"sysfs": src}), flush=True)
    ws = [phant_amd.witness.account_witness(100_000, depth=8, seed=2 + i, device=dev) for i in range(4)]
    streams = [torch.cuda.Stream() for _ in range(4)]
    ctxs = {m: [phant_amd.Context(0, verify_nodedup=(m == "nodedup")) for _ in range(4)] for m in ("flat", "nodedup")}
    status = [torch.empty(w.batch.n, dtype=torch.uint8, device=dev) for w in ws]
    big = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
    rate_ctx = phant_amd.Context(0)

    def verify(mode):
        def f():
            for i in range(4):
                with torch.cuda.stream(streams[i]):
                    M.verify_batch_dev(ws[i].batch, status=status[i], ctx=ctxs[mode][i])
            return 400_000
        return f

    def stream_read():
        big.view(torch.int64).max()
        return big.numel()

    def keccak_only():
        rate_ctx.keccak_rate(4, 400)
        return 0

    phases = [("idle", None), ("keccak_f_only", keccak_only), ("read_stream_1GiB", stream_read), ("verify_flat_4_in_flight", verify("flat")),
              ("verify_nodedup_4_in_flight", verify("nodedup")), ("keccak_f_only_again", keccak_only)]
    lines = []
    for name, fn in phases:
        if fn is not None:
            fn()
            torch.cuda.synchronize()
        s = Sampler(src)
        mid = {}
        probe = threading.Timer(min(1.0, args.seconds / 2), lambda: mid.update(rocm_smi=smi()))   # one reading WHILE the phase runs
        t0 = time.perf_counter()
        s.start()
        probe.start()
        units, calls = 0, 0
        while time.perf_counter() - t0 < args.seconds:
            if fn is None:
                time.sleep(0.05)
            else:
                for _ in range(8):
                    units += fn()
                    calls += 1
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        s.stop_flag = True
        s.join()
        probe.join()
        line = {"phase": name, "seconds": round(dt, 3), "calls": calls, "units_per_s": round(units / dt), **s.summary()}
        if name.startswith("keccak_f_only"):
            line["keccak_f_per_s"] = rate_ctx.keccak_rate(4, 100)
        line.update(mid)
        print(json.dumps(line), flush=True)
        lines.append(line)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            for l in lines:
                f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()
