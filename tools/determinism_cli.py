#!/usr/bin/env python
"""Run-to-run determinism of build/difacto on the rcv1 fixture: every (device_path, store) combination N times, the
per-epoch training losses printed side by side.  usage: determinism_cli.py [N]"""
import os, subprocess, sys, tempfile
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
d = tempfile.mkdtemp(prefix="det_")
base = open(os.path.join(R, "example", "rcv1_fm.conf")).read().replace("V_init = refrand", "V_init = hash")
base = base.replace("max_num_epochs = 10", "max_num_epochs = 3").replace("batch_size = 100", "batch_size = 25")
exe = os.path.join(R, "build", "difacto")
shard_env = dict(os.environ, DMLC_ROLE="worker", DMLC_NUM_WORKER="1", DIFACTO_RANK="0", DIFACTO_DEVICE="0")
for path in ("fused", "literal"):
    conf = os.path.join(d, path + ".conf")
    open(conf, "w").write(base.replace("device_path = fused", "device_path = " + path))
    for store, env in (("plain", os.environ), ("sharded1", shard_env)):
        for i in range(n):
            r = subprocess.run([exe, "argfile=" + conf], capture_output=True, text=True, timeout=600, cwd=R, env=env)
            ls = [l.split("loss = ")[1].split(",")[0] for l in r.stderr.splitlines() if "Training: loss" in l]
            print("%-8s %-9s run %d rc %d losses %s" % (path, store, i, r.returncode, " ".join(ls)), flush=True)
