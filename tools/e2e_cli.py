#!/usr/bin/env python
"""End-to-end rate of build/difacto from files (PCIe and parsing included): Criteo-shaped rows written
as criteo text, libsvm text and .rec (RecordIO of LZ4 compressed row blocks), one training epoch each
with the C3 hyper-parameters.  Every format is run on the generated file (`rows` rows) and on that file
repeated `rep` times; the difference of the two wall times over the difference of the row counts is the
steady-state rate (process start, table allocation and the first-touch costs cancel).
usage: e2e_cli.py [rows [rep]] -> one JSON object per format on stdout"""
import json, os, re, subprocess, sys, tempfile, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
from oracle import ingest as oi

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(prefix="e2e_")
# tokens: 13 integer slots (10 000 values each), 26 categorical slots (8 hex chars, Zipf-ish)
ints = rng.zipf(1.3, size=(rows, 13)) % 10000
cats = (rng.zipf(1.1, size=(rows, 26)) % 1000000).astype(np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)
lab = (rng.random(rows) < 0.25).astype(np.int32)
t0 = time.time()
lines = ["%d\t%s\t%s" % (lab[i], "\t".join(map(str, ints[i])), "\t".join("%08x" % c for c in cats[i])) for i in range(rows)]
text = ("\n".join(lines) + "\n").encode()
open(os.path.join(d, "train.criteo"), "wb").write(text)
off, labf, idx = oi.parse_criteo(text)
with open(os.path.join(d, "train.libsvm"), "w") as f:
    for i in range(rows):
        f.write("%d %s\n" % (lab[i], " ".join("%d:1" % v for v in idx[int(off[i]):int(off[i + 1])])))
recs = []
for a in range(0, rows, 10000):
    b = min(rows, a + 10000)
    recs.append(oi.write_crb_record(off[a:b + 1] - off[a], labf[a:b], idx[int(off[a]):int(off[b])]))
open(os.path.join(d, "train.rec"), "wb").write(oi.write_recordio(recs))
sys.stderr.write("files written in %.1f s\n" % (time.time() - t0))
common = ["task=train", "learner=sgd", "batch_size=" + os.environ.get("E2E_BATCH_SIZE", "10000"), "max_num_epochs=1",
          "V_dim=" + os.environ.get("E2E_VDIM", "64"), "V_threshold=0", "l1=0", "lr=.01",
          "V_lr=.01", "V_init=hash", "table_capacity=" + os.environ.get("E2E_TABLE_CAPACITY", "8388608"), "stop_rel_objv=0", "num_jobs_per_epoch=1"]
EXES = os.environ.get("E2E_EXES", "difacto").split(",")   # A/B: several binaries under build/ on the same files
# A/B of environment switches on the same files: E2E_VARIANTS="name:KEY=VAL+KEY=VAL,name2:" (an entry of EXES may be
# "binary@name" to run that binary under the named variant's environment)
VARIANTS = {}
for item in filter(None, os.environ.get("E2E_VARIANTS", "").split(",")):
    name, _, kv = item.partition(":")
    VARIANTS[name] = dict(x.split("=", 1) for x in kv.split("+") if x)
exe = EXES[0]
def run(path, fmt):
    t0 = time.time()
    binary, _, var = exe.partition("@")
    env = dict(os.environ, **VARIANTS.get(var, {}))
    r = subprocess.run([os.path.join(R, "build", binary), "data_in=" + path, "data_format=" + fmt] + common,
                       capture_output=True, text=True, timeout=900, env=env)
    dt = time.time() - t0
    loss = [l for l in r.stderr.splitlines() if "Training: loss" in l]
    loop_s = None
    for l in r.stderr.splitlines():
        if "host loop over" in l or "reader: " in l or "batch reader" in l or "start-up:" in l or "dfh_table:" in l or "process:" in l:   # DIFACTO_PROFILE=1
            sys.stderr.write(fmt + " " + exe + ": " + l.split("INFO")[-1].strip() + "\n")
        m = re.search(r"host loop over (\d+) minibatches: reader ([0-9.e+-]+) s, stage \+ localize \+ lookup ([0-9.e+-]+) s.*step ([0-9.e+-]+) s", l)
        if m:   # the worker loop's own clock: process start, HIP initialisation and the table allocation are outside it
            loop_s = float(m.group(2)) + float(m.group(3)) + float(m.group(4))
    return dt, r.returncode, (loss[-1].split("INFO")[-1].strip() if loss else r.stderr[-300:]), loop_s


for fmt in os.environ.get("E2E_FORMATS", "criteo,libsvm,rec").split(","):
    path = os.path.join(d, "train." + fmt)
    big = os.path.join(d, "train_x%d.%s" % (rep, fmt))
    with open(big, "wb") as out:   # text lines and RecordIO records both concatenate
        blob = open(path, "rb").read()
        for _ in range(rep):
            out.write(blob)
    for exe in EXES:
        # the faster of two runs each: the difference of two wall times is sensitive to a hiccup in either
        runs1 = [run(path, fmt) for _ in range(3)]
        runs2 = [run(big, fmt) for _ in range(3)]
        dt1, rc1, line1, _ = min(runs1)
        dt2, rc2, line2, _ = min(runs2)
        steady = rows * (rep - 1) / max(dt2 - dt1, 1e-9)
        # the same difference by the worker loop's own clock (DIFACTO_PROFILE=1): the wall times carry ~0.4 s of process
        # start whose run-to-run spread (+-0.1 s) is as large as the epoch itself
        l1 = [x[3] for x in runs1 if x[3] is not None]
        l2 = [x[3] for x in runs2 if x[3] is not None]
        loop = dict(loop_s=min(l1), loop_s_big=min(l2), loop_rows_per_s_big=rows * rep / min(l2),
                    steady_rows_per_s_by_loop_clock=rows * (rep - 1) / max(min(l2) - min(l1), 1e-9)) if l1 and l2 else {}
        print(json.dumps(dict(format=fmt, exe=exe, rows=rows, file_mb=os.path.getsize(path) / 1e6, wall_s=dt1, rows_per_s=rows / dt1,
                              rc=rc1, line=line1, rows_big=rows * rep, wall_s_big=dt2, rows_per_s_big=rows * rep / dt2, rc_big=rc2,
                              steady_rows_per_s=steady, steady_mb_per_s=steady * os.path.getsize(path) / rows / 1e6, line_big=line2, **loop)),
              flush=True)
    os.remove(big)
