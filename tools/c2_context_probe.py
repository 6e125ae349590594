#!/usr/bin/env python
"""Why does secondary["c2"] of the default bench line read 2.3 M where `bench.py --preset c2` alone reads 3.0 M on the same box?
Runs the c2 line as a child of parents in different states."""
import json, os, subprocess, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
CMD = [sys.executable, os.path.join(R, "bench.py"), "--preset", "c2", "--no-secondary", "--min-time", "1", "--cpu-batches", "0"]


def child(label, extra=(), env=None):
    r = subprocess.run(CMD + list(extra), capture_output=True, text=True, env=env)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print("%-58s %.3f M  median %.1f us  min %.1f max %.1f  enqueue %.1f us" % (label, d["value"] / 1e6, d["ms_per_step"] * 1e3, d["ms_per_step_min"] * 1e3,
                                                                          d["ms_per_step_max"] * 1e3, d["host_enqueue_ms_per_step"] * 1e3), flush=True)


child("parent: plain python")
import torch
child("parent: torch imported")
torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
child("parent: torch has a device context")
from difacto_amd import capi
ctx = capi.Context(0)
tb = capi.Table(ctx, 1 << 20, V_dim=8)
child("parent: + a dfh context and a table alive")
tb.close(); ctx.close(); torch.cuda.empty_cache()
child("parent: context closed again")
child("parent: context closed, child with --cpu-batches 100", extra=["--cpu-batches", "100"])
