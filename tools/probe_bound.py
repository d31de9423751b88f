#!/usr/bin/env python3
"""Exploration: phant_verify_bound_experiment on BASELINE config 3 (what the chip overlaps at best)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import phant_amd
dev = torch.device("cuda", 0)
ctx = phant_amd.Context(0)
# (the tool's own environment: how much the read stream keeps in flight / the region its index wraps in -- handed to the ctx)
for env, knob in (("STREAM_WGS", "stream_wgs"), ("STREAM_MB", "stream_mb")):
    if os.environ.get(env):
        ctx.diag_set(knob, int(os.environ[env]))
wa = phant_amd.witness.account_witness(int(os.environ.get("PROOFS", "100000")), depth=8, seed=2, device=dev, ctx=ctx)
for k in range(3):
    print(ctx.verify_form(), ctx.verify_bound_experiment(wa.batch, 20), flush=True)
