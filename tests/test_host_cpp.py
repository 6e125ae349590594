"""The C++ host side (include/difacto/*.h over the C ABI): builds on CPU, runs on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "golden", "rcv1_100.libsvm")


@pytest.fixture(scope="module")
def built():
    from difacto_amd import build
    build.build_hip()
    build.build_host()
    return os.path.join(ROOT, "build")


def test_host_binaries_build(built):
    for exe in ("difacto", "difacto_host_tests"):
        assert os.access(os.path.join(built, exe), os.X_OK)
    # the CLI prints usage without arguments, like the reference (src/main.cc:39-42)
    r = subprocess.run([os.path.join(built, "difacto")], capture_output=True, text=True, timeout=60)
    assert "usage: difacto key1=val1" in r.stderr


def test_batch_reader_checksums_on_cpu(built):
    """the reader that feeds the path against the reference's own minibatch checksums
    (tests/cpp/batch_reader_test.cc:9-57: Read, RandRead, PartRead) - host only, no device touched"""
    r = subprocess.run([os.path.join(built, "difacto_host_tests"), DATA, "reader"], capture_output=True, text=True, timeout=120)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and "ALL HOST TESTS PASSED" in r.stdout


def test_cli_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([os.path.join(built, "difacto"), "data_in=" + DATA, "batch_size=100", "V_dim=0"],
                       capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "dfh_ctx_create" in r.stderr  # no CPU fallback


@pytest.mark.gpu
def test_host_interface_tests_on_gpu(built):
    """FMLoss.NoV/HasV, Localizer.*, SGDLearner.Basic (fused + literal), Store/model I/O through the C++ interfaces"""
    r = subprocess.run([os.path.join(built, "difacto_host_tests"), DATA], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "ALL HOST TESTS PASSED" in r.stdout


@pytest.mark.gpu
def test_cli_trains_from_conf(built, tmp_path):
    model = os.path.join(tmp_path, "model.bin")
    r = subprocess.run([os.path.join(built, "difacto"), "argfile=" + os.path.join(ROOT, "example", "rcv1_fm.conf"),
                        "model_out=" + model, "bogus_key=1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    print(r.stderr[-3000:])
    assert r.returncode == 0
    assert "Start epoch 0" in r.stderr and "Validation: loss = " in r.stderr and "AUC = " in r.stderr
    assert "Unrecognized keyword argument" in r.stderr and "bogus_key" in r.stderr  # src/main.cc:25-31
    losses = [float(l.split("loss = ")[1].split(",")[0]) for l in r.stderr.splitlines() if "Training: loss" in l]
    assert len(losses) >= 2 and losses[-1] < losses[0]
    assert os.path.getsize(model) > 100
    # the file the C++ host wrote loads through the C ABI (one format, two writers/readers)
    from difacto_amd import capi
    ctx = capi.Context(0)
    tb = capi.Table(ctx, 1 << 17, V_dim=8, init_mode=capi.INIT_REFRAND)
    n, _ = tb.load(model)
    assert n > 1000 and tb.size() == n
    tb.close()
    ctx.close()
    # resume from the saved model: the first epoch starts where the last one ended
    r2 = subprocess.run([os.path.join(built, "difacto"), "argfile=" + os.path.join(ROOT, "example", "rcv1_fm.conf"),
                         "model_in=" + model, "max_num_epochs=1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r2.returncode == 0 and "model loaded from" in r2.stderr
    first = [float(l.split("loss = ")[1].split(",")[0]) for l in r2.stderr.splitlines() if "Training: loss" in l][0]
    assert first < losses[0]


def _hash_conf(tmp_path, epochs, batch_size=100):
    """example/rcv1_fm.conf with the order-independent V init the sharded store needs (the text of an
    argfile is parsed after the command line, so its keys win: edit the file, not the arguments)"""
    text = open(os.path.join(ROOT, "example", "rcv1_fm.conf")).read().replace("V_init = refrand", "V_init = hash")
    text = text.replace("max_num_epochs = 10", "max_num_epochs = %d" % epochs).replace("batch_size = 100", "batch_size = %d" % batch_size)
    assert "V_init = hash" in text and "max_num_epochs = %d" % epochs in text
    path = os.path.join(tmp_path, "rcv1_fm_hash_%d_%d.conf" % (epochs, batch_size))
    open(path, "w").write(text)
    return path


def _losses(stderr):
    return [float(l.split("loss = ")[1].split(",")[0]) for l in stderr.splitlines() if "Training: loss" in l]


@pytest.mark.gpu
def test_cli_sharded_store_one_rank_matches_plain_run(built, tmp_path):
    """DMLC_ROLE selects the sharded store (Store::Create, factories.cc): with one rank over RCCL the
    worker loop through dfh_shard_step must train exactly like the plain fused loop"""
    args = [os.path.join(built, "difacto"), "argfile=" + _hash_conf(tmp_path, 4)]
    plain = subprocess.run(args, capture_output=True, text=True, timeout=600, cwd=ROOT)
    env = dict(os.environ, DMLC_ROLE="worker", DMLC_NUM_WORKER="1", DIFACTO_RANK="0", DIFACTO_DEVICE="0")
    shard = subprocess.run(args + ["model_out=" + os.path.join(tmp_path, "m")], capture_output=True, text=True, timeout=600,
                           cwd=ROOT, env=env)
    print(shard.stderr[-3000:])
    assert plain.returncode == 0 and shard.returncode == 0
    a, b = _losses(plain.stderr), _losses(shard.stderr)
    assert len(a) == len(b) and len(a) >= 2  # the validation-AUC criterion may stop both runs early
    assert all(abs(x - y) <= 2e-5 * abs(x) for x, y in zip(a, b)), (a, b)  # same minibatches (RefRand), same kernels
    assert os.path.getsize(os.path.join(tmp_path, "m.part-0")) > 100


@pytest.mark.gpu
def test_cli_two_ranks_share_the_gpu_over_files(built, tmp_path):
    """build/difacto as TWO processes (DMLC_NUM_WORKER=2) on the one GPU of the test box, the exchange of
    dfh_shard_step carried by the file transport (RCCL cannot put two ranks on one device): both ranks
    must report the same merged progress, the loss must fall, every rank writes its model part, and a
    single process can load the union of the parts"""
    rv = os.path.join(tmp_path, "rv")
    os.makedirs(rv)
    model = os.path.join(tmp_path, "model")
    args = [os.path.join(built, "difacto"), "argfile=" + _hash_conf(tmp_path, 5, 25), "model_out=" + model]
    open(model + ".part-2", "wb").write(b"stale part of an earlier save with three ranks")   # must not survive
    procs = []
    for r in range(2):
        env = dict(os.environ, DMLC_ROLE="worker", DMLC_NUM_WORKER="2", DIFACTO_RANK=str(r), DIFACTO_DEVICE="0",
                   DIFACTO_COMM="file", DIFACTO_RENDEZVOUS=rv)
        procs.append(subprocess.Popen(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=120) for p in procs]
    for p, (_, err) in zip(procs, outs):
        print(err[-2500:])
        assert p.returncode == 0
    l0, l1 = _losses(outs[0][1]), _losses(outs[1][1])
    assert len(l0) >= 2 and l0 == l1, (l0, l1)          # the merged record is identical on both ranks
    assert l0[-1] < l0[0]
    assert not os.path.exists(model + ".part-2") and open(model + ".parts").read().split() == ["2"]
    from difacto_amd import capi
    ctx = capi.Context(0)
    tb = capi.Table(ctx, 1 << 17, V_dim=8, init_mode=capi.INIT_HASH)
    n = sum(tb.load(model + ".part-%d" % r)[0] for r in range(2))
    assert n > 1000 and tb.size() == n
    tb.close()
    ctx.close()


@pytest.mark.gpu
def test_cli_trains_from_criteo_text_and_rec(built, tmp_path):
    """data_format = criteo (CityHash64 + slot tag) and data_format = rec (RecordIO of LZ4 compressed row
    blocks): the formats of the reference's example/criteo_sgd.conf — and data_format = adfea (`lineid count label
    idx:gid ...`, src/reader/adfea_parser.h).  The same rows in the three formats must give the same training trajectory."""
    import numpy as np
    from oracle import ingest as oi
    from test_ingest import _criteo_text
    rng = np.random.default_rng(21)
    text = _criteo_text(rng, 3000)
    txt = os.path.join(tmp_path, "train.criteo")
    open(txt, "wb").write(text)
    off, lab, idx = oi.parse_criteo(text)
    recs = []
    for a in range(0, 3000, 500):   # six compressed row blocks of 500 rows
        o = off[a:a + 501] - off[a]
        recs.append(oi.write_crb_record(o, lab[a:a + 500], idx[int(off[a]):int(off[a + 500])]))
    rec = os.path.join(tmp_path, "train.rec")
    open(rec, "wb").write(oi.write_recordio(recs))
    assert set(np.unique(lab)) <= {0.0, 1.0}
    adf = os.path.join(tmp_path, "train.adfea")
    with open(adf, "w") as f:   # id = EncodeFeaGrpID(id >> 12, id & 0xfff, 12): the same 64-bit ids
        for r in range(3000):
            ids = idx[int(off[r]):int(off[r + 1])]
            f.write(" ".join(["%d" % r, "%d" % len(ids), "%d" % lab[r]] + ["%d:%d" % (int(v) >> 12, int(v) & 0xFFF) for v in ids]) + "\n")
    common = ["task=train", "learner=sgd", "batch_size=500", "max_num_epochs=3", "V_dim=4", "V_threshold=0", "l1=.01", "lr=.1",
              "V_lr=.05", "V_init=hash", "table_capacity=262144", "stop_rel_objv=0",
              "num_jobs_per_epoch=1",   # one data part: byte-range parts of a text file and of a RecordIO file hold different rows
              "shuffle=0"]              # file order: the two readers cut the stream into different shuffle buffers
    runs = []
    for fmt, path in (("criteo", txt), ("rec", rec), ("adfea", adf)):
        r = subprocess.run([os.path.join(built, "difacto"), "data_in=" + path, "data_format=" + fmt] + common,
                           capture_output=True, text=True, timeout=180, cwd=ROOT)
        print(r.stderr[-1500:])
        assert r.returncode == 0
        runs.append(_losses(r.stderr))
    assert len(runs[0]) == 3 and runs[0] == runs[1] == runs[2], runs
    assert runs[0][-1] < runs[0][0]


@pytest.mark.gpu
def test_cli_reference_criteo_conf_unchanged_on_a_model_that_outgrows_the_first_allocation(built, tmp_path):
    """the reference's own example/criteo_sgd.conf, key for key (only the two file names differ): no table_capacity,
    no device-only key.  The data brings ~3 M distinct ids (every categorical token random): the model table starts at
    2^20 rows and must grow — the reference's unordered_map is unbounded (sgd_updater.h:78) — where round 3's fixed
    default of 2^22 rows... and any fixed default... would end a bigger file in DFH_ERR_CAPACITY."""
    import numpy as np
    from oracle import ingest as oi
    rng = np.random.default_rng(5)
    nrows, blk = 110000, 10000
    lab = (rng.random(nrows) < 0.25).astype(np.float32)
    # 13 small-vocabulary slots + 26 slots of random 52-bit tokens, slot id in the low 12 bits (EncodeFeaGrpID, base.h:60-63)
    tok = rng.integers(0, 1 << 52, size=(nrows, 39), dtype=np.uint64)
    tok[:, :13] = rng.integers(0, 5000, size=(nrows, 13), dtype=np.uint64)
    idx = ((tok << np.uint64(12)) | np.arange(39, dtype=np.uint64)[None, :]).reshape(-1)
    recs = []
    for a in range(0, nrows, blk):
        n = min(blk, nrows - a)
        o = (np.arange(n + 1) * 39).astype(np.uint64)
        recs.append(oi.write_crb_record(o, lab[a:a + n], idx[a * 39:(a + n) * 39]))
    rec = os.path.join(tmp_path, "criteo_train.rec")
    open(rec, "wb").write(oi.write_recordio(recs))
    ref_conf = """# data
data_in = %s
data_val = %s
data_format = rec

# learner
task = train
learner = sgd
max_num_epochs = 10
batch_size = 10000

# linear term
l1 = 10
l2 = 10

# embedding term
V_dim = 10
V_threshold = 10
V_l2 = 10
""" % (rec, rec)
    conf = os.path.join(tmp_path, "criteo_sgd.conf")
    open(conf, "w").write(ref_conf)
    r = subprocess.run([os.path.join(built, "difacto"), "argfile=" + conf], capture_output=True, text=True, timeout=900, cwd=ROOT)
    print(r.stderr[-2500:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Start epoch 0" in r.stderr and "Validation: loss = " in r.stderr
    assert "model table is full" not in r.stderr and "Unrecognized keyword" not in r.stderr


@pytest.mark.gpu
def test_cli_runs_are_reproducible(built, tmp_path):
    """two runs of one command give the same trajectory, shuffle buffer included: the permutation comes from the reader's
    own restatement of the reference's rand() stream (RefRand, batch_reader.h), not from the process-wide rand(), which
    the HIP runtime's threads draw from as well; every device kernel on the path is deterministic"""
    exe = os.path.join(built, "difacto")
    for path in ("fused", "literal"):
        text = open(_hash_conf(tmp_path, 3, 25)).read().replace("device_path = fused", "device_path = " + path)
        conf = os.path.join(tmp_path, "repro_%s.conf" % path)
        open(conf, "w").write(text)
        lines = []
        for _ in range(3):
            r = subprocess.run([exe, "argfile=" + conf], capture_output=True, text=True, timeout=600, cwd=ROOT)
            assert r.returncode == 0, r.stderr[-1500:]
            lines.append([l.split("] ")[-1] for l in r.stderr.splitlines() if "loss = " in l])
        assert len(lines[0]) >= 3 and lines[0] == lines[1] == lines[2], (path, lines)


@pytest.mark.gpu
def test_cli_device_feed_matches_host_feed(built, tmp_path):
    """training with a shuffle buffer reads through the device feed (shuffle buffers uploaded once, minibatches gathered on
    the device by row number); DIFACTO_HOST_FEED=1 keeps the host-side gather.  Same minibatches, so the same losses —
    several buffers, down-sampling (minibatches that straddle two buffers), values and binary data"""
    import numpy as np
    from test_ingest import _criteo_text
    exe = os.path.join(built, "difacto")
    rng = np.random.default_rng(33)
    txt = os.path.join(tmp_path, "train.criteo")
    open(txt, "wb").write(_criteo_text(rng, 4000))
    sparse = os.path.join(tmp_path, "sparse.criteo")
    lines = []
    for _ in range(12000):
        f = [str(int(rng.random() < 0.03))] + [str(int(rng.integers(0, 50))) for _ in range(13)] + ["%08x" % int(rng.integers(0, 1 << 20)) for _ in range(26)]
        lines.append("\t".join(f))
    open(sparse, "w").write("\n".join(lines) + "\n")
    runs = {}
    cases = [("criteo", ["data_in=" + txt, "data_format=criteo", "batch_size=300", "shuffle=2", "neg_sampling=0.7", "V_dim=4", "V_threshold=0",
                         "l1=.01", "lr=.1", "V_lr=.05", "V_init=hash", "table_capacity=262144", "max_num_epochs=3", "stop_rel_objv=0",
                         "num_jobs_per_epoch=2"]),
             ("rcv1", ["argfile=" + _hash_conf(tmp_path, 3, 25), "shuffle=2"]),
             # one shuffle buffer per minibatch and 97 % of the negatives dropped on data with 3 % positives: a
             # minibatch draws its rows from ~17 buffers, far more than the six-slot ring of round 3 held
             ("sparse", ["data_in=" + sparse, "data_format=criteo", "batch_size=200", "shuffle=1", "neg_sampling=0.97", "V_dim=4",
                         "V_threshold=0", "l1=.01", "lr=.1", "V_lr=.05", "V_init=hash", "max_num_epochs=2", "stop_rel_objv=0",
                         "num_jobs_per_epoch=1"])]
    for name, args in cases:
        for feed in ("device", "host"):
            env = dict(os.environ)
            if feed == "host":
                env["DIFACTO_HOST_FEED"] = "1"
            r = subprocess.run([exe, "task=train", "learner=sgd"] + args, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
            assert r.returncode == 0, r.stderr[-2000:]
            runs[name, feed] = [l.split("] ")[-1] for l in r.stderr.splitlines() if "loss = " in l]
        assert len(runs[name, "device"]) >= 2 and runs[name, "device"] == runs[name, "host"], (name, runs[name, "device"], runs[name, "host"])


@pytest.mark.gpu
def test_cli_task_predict_matches_the_oracle(built, tmp_path):
    """task=predict (a TODO in the reference's src/main.cc:61-62; sgd_param.h:24-28 names model_in for it): train with
    model_out, then a second process loads model_in and writes one logit per example to pred_out — several data parts
    and several minibatches per part, in file order.  Checked against FMLoss::Predict of the oracle on the weights of
    the saved model (rtol 1e-5 + the summation floor of oracle/tolerance.py); pred_prob=1 gives the sigmoid."""
    import numpy as np
    from conftest import load_libsvm
    from difacto_amd import capi
    from oracle import bindings as ob, tolerance as T
    model = os.path.join(tmp_path, "model.bin")
    exe = os.path.join(built, "difacto")
    r = subprocess.run([exe, "argfile=" + os.path.join(ROOT, "example", "rcv1_fm.conf"), "model_out=" + model],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    conf = os.path.join(tmp_path, "predict.conf")
    text = open(os.path.join(ROOT, "example", "rcv1_predict.conf")).read().replace("batch_size = 100", "batch_size = 13")
    open(conf, "w").write(text + "\nnum_jobs_per_epoch = 3\n")
    pred = os.path.join(tmp_path, "pred.txt")
    r = subprocess.run([exe, "argfile=" + conf, "model_in=" + model, "pred_out=" + pred], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    print(r.stderr[-2000:])
    assert r.returncode == 0 and "predicted 100 examples" in r.stderr
    got = np.loadtxt(pred, dtype=np.float32)
    assert got.shape == (100,)
    # the oracle's FMLoss::Predict on the saved model's weights
    off, idx, val, lab = load_libsvm(DATA)
    O = ob.Oracle()
    loc = O.localize(off, idx)
    ctx = capi.Context(0)
    tb = capi.Table(ctx, 1 << 17, V_dim=8, init_mode=capi.INIT_REFRAND)
    tb.load(model)
    vals, lens = tb.pull(loc["feaids"])
    tb.close()
    ctx.close()
    wp, vp = O.get_pos(lens)
    want = O.fm_predict(8, loc["offset"], loc["index"], val, vals, wp, vp)
    w64, V64, _ = T.dense_rows(vals, lens, 8)
    _, floor = T.predict_bound(T.design(loc["offset"], loc["index"], val, loc["U"]), w64, V64)
    T.check(got, want, floor + 1e-7 * np.abs(want), "task=predict logits vs FMLoss::Predict")  # + the %.9g text round trip
    assert np.abs(want).max() > 0.05   # a trained model, not zeros
    # probabilities
    r = subprocess.run([exe, "argfile=" + conf, "model_in=" + model, "pred_out=" + pred, "pred_prob=1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0
    prob = np.loadtxt(pred, dtype=np.float32)
    np.testing.assert_allclose(prob, 1.0 / (1.0 + np.exp(-got.astype(np.float64))), rtol=2e-6)
    # a prediction task without a model is refused (sgd_param.h:24-28)
    r = subprocess.run([exe, "argfile=" + conf, "pred_out=" + pred], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "needs model_in" in r.stderr


@pytest.mark.gpu
def test_cli_refuses_roles_other_than_worker_and_a_missing_rank(built, tmp_path):
    """a dmlc launch starts a scheduler and servers beside the workers; here every process with DMLC_ROLE set
    would have joined as rank 0.  Other roles and DMLC_NUM_WORKER > 1 without DIFACTO_RANK are refused with a message."""
    args = [os.path.join(built, "difacto"), "argfile=" + _hash_conf(tmp_path, 1)]
    for env_add, msg in ((dict(DMLC_ROLE="scheduler", DMLC_NUM_WORKER="2"), "runs workers only"),
                         (dict(DMLC_ROLE="server", DMLC_NUM_WORKER="1", DIFACTO_RANK="0"), "runs workers only"),
                         (dict(DMLC_ROLE="worker", DMLC_NUM_WORKER="2"), "needs DIFACTO_RANK")):
        env = {k: v for k, v in os.environ.items() if k not in ("DIFACTO_RANK", "DMLC_ROLE", "DMLC_NUM_WORKER")}
        env.update(env_add, DIFACTO_DEVICE="0")
        r = subprocess.run(args, capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert r.returncode != 0 and msg in r.stderr, (env_add, r.stderr[-800:])


@pytest.mark.gpu
def test_cli_literal_path_on_the_sharded_store(built, tmp_path):
    """device_path=literal on the sharded store: the worker loop makes the reference's own calls (Push(kFeaCount), Pull,
    Loss::Predict / CalcGrad, Push(kGradient)) and the store routes them to the owners (dfh_shard_push_host /
    dfh_shard_pull_host).  One rank: the trajectory of the plain literal run.  Two ranks over the file transport: both
    report the same merged progress and the loss falls."""
    text = open(_hash_conf(tmp_path, 3, 25)).read().replace("device_path = fused", "device_path = literal")
    conf = os.path.join(tmp_path, "literal.conf")
    open(conf, "w").write(text)
    exe = os.path.join(built, "difacto")
    plain = subprocess.run([exe, "argfile=" + conf], capture_output=True, text=True, timeout=600, cwd=ROOT)
    env = dict(os.environ, DMLC_ROLE="worker", DMLC_NUM_WORKER="1", DIFACTO_RANK="0", DIFACTO_DEVICE="0")
    one = subprocess.run([exe, "argfile=" + conf], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    print(one.stderr[-1500:])
    assert plain.returncode == 0 and one.returncode == 0
    a, b = _losses(plain.stderr), _losses(one.stderr)
    # two different kernels apply the pushes (k_push_grad: a wave per key; k_push_grad_multi: a lane group per key): the
    # trajectories agree to accumulated rounding, not bit for bit.  (Before the reader had its own rand() stream the two
    # processes drew different shuffles and this comparison needed 1e-3.)
    assert len(a) == len(b) >= 2 and all(abs(x - y) <= 2e-5 * abs(x) for x, y in zip(a, b)), (a, b)
    rv = os.path.join(tmp_path, "rv_lit")
    os.makedirs(rv)
    procs = []
    for r in range(2):
        env = dict(os.environ, DMLC_ROLE="worker", DMLC_NUM_WORKER="2", DIFACTO_RANK=str(r), DIFACTO_DEVICE="0", DIFACTO_COMM="file",
                   DIFACTO_RENDEZVOUS=rv)
        procs.append(subprocess.Popen([exe, "argfile=" + conf], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, cwd=ROOT, env=env))
    outs = [p.communicate(timeout=180) for p in procs]
    for p, (_, err) in zip(procs, outs):
        print(err[-1500:])
        assert p.returncode == 0
    l0, l1 = _losses(outs[0][1]), _losses(outs[1][1])
    assert len(l0) >= 2 and l0 == l1 and l0[-1] < l0[0], (l0, l1)
