"""oracle/make_golden.py -- TEST INFRASTRUCTURE ONLY; runs in the build container only.

Produces tests/golden/*.npz by running the REAL reference implementation, imported unmodified
from /root/reference (see oracle/ref_import.py), on seeded inputs:

* retrieval_*.npz     inputs + the ranking returned by the reference's
                      ``evaluate_retrieval.pairwise_retrieval`` (evaluate_retrieval.py:22-73)
* embeddings.npz      the class-embedding matrices shipped by the reference
                      (embeddings/*.pickle: data files, not source) used as loss/metric inputs
* hierarchy_cifar.npz ``ClassHierarchy.hierarchical_precision`` outputs
                      (class_hierarchy.py:211-316) for a small retrieval problem + the taxonomy
                      edges (Cifar-Hierarchy/cifar.parent-child.txt: data file)
* hierarchy_cub.npz, hierarchy_ilsvrc.npz   the same for the reference's CUB balanced (200 classes) and ILSVRC WordNet min-tree
                      (1000 string-id classes) taxonomies
* loss_cifar100.npz   seeded inputs + loss oracle outputs (round-1 fixture, kept)
* loss_ref_*.npz      seeded inputs + the outputs of the reference's OWN utils.py:34-127 and
                      learn_labelembedding.py:17-37, imported unmodified and evaluated on the NumPy
                      ``keras.backend`` stand-in of oracle/keras_stub.py, in float32 (the reference's
                      precision) and float64 (tight values for the oracle)
* lr_schedules.npz    learning-rate trajectories of the reference's get_lr_schedule (utils.py:288-399)
                      with its own clr_callback.py / sgdr_callback.py driven epoch by epoch
* retrieval_d555_*.npz, retrieval_d1000_*.npz   D > 448: this host's OpenBLAS restarts its FMA chain per
                      K block; the probed block list travels in the fixture (``kblocks``)
* topk_head_*.npz   heads (first 256 entries) of the reference's rankings on 640-item D = 555 / D = 1000 problems, for
                      the top-k / sharded-gallery path (features rebuilt from a seed, SHA-1 checked)
* cifar_pipeline.npz   the reference's in-memory data path (datasets/cifar.py, datasets/common.py TinyDatasetGenerator /
                      DataSequence) on a synthetic CIFAR-100 pickle pair: statistics, batches, augmented images + drawn parameters
* imagenet_mintree_unitsphere.npz   the class embedding missing from the reference checkout, regenerated
                      by the reference's compute_class_embedding.py:14-40,176-250 in JSON class order

Usage:  python -m oracle.make_golden        (from the repo root)
"""
import os
import pickle
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import loss_oracle, ref_import  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
REF = ref_import.REFERENCE_ROOT


def load_embedding(name):
    with open(os.path.join(REF, "embeddings", name + ".pickle"), "rb") as f:
        e = pickle.load(f)
    return np.asarray(e["embedding"]), np.asarray(e["ind2label"])


def ref_ranking(er, features, normalize):
    """Run the reference on a COPY (it normalises in place) and return the [N,N] id matrix."""
    ids = list(features.keys()) if isinstance(features, dict) else list(range(len(features)))
    inp = {k: v.copy() for k, v in features.items()} if isinstance(features, dict) else features.copy()
    ret = er.pairwise_retrieval(inp, normalize=normalize, return_generator=False)
    return np.array([ret[i] for i in ids], dtype=np.int32)


def openblas_kblocks(d, q=448):
    """K blocking of OpenBLAS level-3 drivers (driver/level3/level3.c, level3_syrk.c: GEMM_Q = 448 on this host's kernels):
    blocks of q while >= 2q remain, then the rest split in two halves (rounded up) if it exceeds q."""
    out, ls = [], 0
    while ls < d:
        m = d - ls
        if m >= 2 * q:
            m = q
        elif m > q:
            m = (m + 1) // 2
        out.append(m)
        ls += m
    return out


def probe_kblocks(feat):
    """The K-block list under which the canonical FMA chain reproduces THIS host's BLAS bit for bit (checked, not assumed)."""
    from oracle import retrieval_oracle as ro
    x = np.ascontiguousarray(feat[:96], dtype=np.float32)
    want = np.dot(x, x.T)
    for kb in ([x.shape[1]], openblas_kblocks(x.shape[1])):
        if np.array_equal(want, ro.canon_pdist(x, None, ro.METRIC_DOT, kblocks=kb)):
            return kb
    raise RuntimeError("host BLAS summation order not reproduced for D=%d" % x.shape[1])



def hierarchy_goldens_more(er, ch):
    """hierarchy_cub.npz / hierarchy_ilsvrc.npz: ``ClassHierarchy.hierarchical_precision`` (class_hierarchy.py:211-316) of the
    unmodified reference on its own rankings for the two other taxonomies it ships -- CUB balanced (200 integer classes) and the
    ILSVRC WordNet min-tree (1000 string-id classes: more than 256, the 16-bit class table of the GPU kernel).  Small retrieval
    problems (the reference walks every ranking in Python); whole-list and clipped AHP, AP, cosine and Euclidean branch."""
    import json
    rng = np.random.default_rng(23)
    cases = {}
    with open(os.path.join(REF, "CUB-Hierarchy", "classes_balanced.txt")) as f:
        cub_classes = [int(l.split()[0]) for l in f if l.strip()]
    cases["cub"] = (os.path.join(REF, "CUB-Hierarchy", "cub_balanced.parent-child.txt"), int, cub_classes[:200], 520, 24)
    with open(os.path.join(REF, "ILSVRC", "imagenet_class_index.json")) as f:
        idx = json.load(f)
    cases["ilsvrc"] = (os.path.join(REF, "ILSVRC", "wordnet.parent-child.mintree.txt"), str, [idx[str(i)][0] for i in range(1000)], 1400, 32)
    for name, (path, id_type, classes, n, d) in cases.items():
        hier = ch.ClassHierarchy.from_file(path, id_type=id_type)
        with open(path) as f:
            edges = np.array([l.split()[:2] for l in f if l.strip()])
        lab_idx = np.concatenate([np.arange(len(classes)), rng.integers(0, len(classes), size=n - len(classes))]) if n > len(classes) \
            else rng.integers(0, len(classes), size=n)
        rng.shuffle(lab_idx)
        labels = [classes[i] for i in lab_idx]
        centers = rng.standard_normal((len(classes), d)).astype(np.float32)
        feats = (centers[lab_idx] + 0.9 * rng.standard_normal((n, d))).astype(np.float32)
        ks = [1, 10, 50, 100]
        res = {}
        for norm in (True, False):
            for ahp in (True, 50):
                avg, _ = hier.hierarchical_precision(er.pairwise_retrieval(feats.copy(), normalize=norm), labels, ks, compute_ahp=ahp,
                                                     compute_ap=True, all_ids=list(range(n)))
                for m, v in avg.items():
                    res["%s|norm=%d|ahp=%s" % (m, norm, ahp)] = v
        np.savez_compressed(os.path.join(OUT, "hierarchy_%s.npz" % name), edges=edges, labels=np.array(labels), features=feats,
                            ks=np.array(ks), metric_names=np.array(list(res.keys())), metric_values=np.array(list(res.values())),
                            id_type=np.array("int" if id_type is int else "str"))
        print("hierarchy", name, "classes", len(set(labels)), "n", n, {k: round(v, 4) for k, v in list(res.items())[:3]})

def regenerate_imagenet_mintree():
    """embeddings/imagenet_mintree.unitsphere.pickle is missing from the checkout (.MISSING_LARGE_BLOBS): rerun the reference's
    compute_class_embedding.py (unitsphere method) on ILSVRC/wordnet.parent-child.mintree.txt with the class list in the order
    of ILSVRC/imagenet_class_index.unitsphere.json."""
    import json
    import subprocess
    import tempfile
    with open(os.path.join(REF, "ILSVRC", "imagenet_class_index.unitsphere.json")) as f:
        idx = json.load(f)
    synsets = [idx[str(i)][0] for i in range(len(idx))]
    with tempfile.TemporaryDirectory() as tmp:
        cl = os.path.join(tmp, "classes.txt")
        with open(cl, "w") as f:
            f.write("\n".join(synsets) + "\n")
        out = os.path.join(tmp, "imagenet_mintree.unitsphere.pickle")
        env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
        subprocess.check_call([sys.executable, os.path.join(REF, "compute_class_embedding.py"), "--hierarchy",
                               os.path.join(REF, "ILSVRC", "wordnet.parent-child.mintree.txt"), "--str_ids", "--class_list", cl,
                               "--out", out], env=env, cwd=tmp)
        with open(out, "rb") as f:
            e = pickle.load(f)
    emb = np.asarray(e["embedding"])
    assert emb.shape == (1000, 1000) and list(e["ind2label"]) == synsets
    assert np.abs(np.linalg.norm(emb, axis=1) - 1).max() < 1e-9
    np.savez_compressed(os.path.join(OUT, "imagenet_mintree_unitsphere.npz"), embedding=emb.astype(np.float32),
                        ind2label=np.array(synsets), max_abs_f64_minus_f32=np.abs(emb - emb.astype(np.float32)).max())
    print("imagenet_mintree.unitsphere", emb.shape)
    return emb


def loss_reference_goldens(emb):
    """Evaluate the reference's own loss / metric source lines (imported unmodified) on seeded inputs."""
    refs = {fx: (ref_import.import_reference("utils", fx), ref_import.import_reference("learn_labelembedding", fx))
            for fx in ("float32", "float64")}
    for key, B, noise in (("cifar100_unitsphere", 128, 0.35), ("cifar100_glove", 96, 0.35), ("nab_sim8", 96, 0.35),
                          ("cub_balanced_unitsphere", 64, 0.35), ("imagenet_mintree_unitsphere", 48, 0.05)):
        E = np.asarray(emb[key], dtype=np.float64)
        C, D = E.shape
        rng = np.random.default_rng(C * 7 + D)
        y = rng.integers(0, C, size=B)
        if key == "nab_sim8":
            zero = np.nonzero(np.abs(E).sum(axis=1) == 0)[0]
            y[:len(zero[:8])] = zero[:8]                     # classes whose embedding is the zero vector
        x = (E[y] + noise * rng.standard_normal((B, D)) * np.abs(E).mean()).astype(np.float32)
        x[::3] *= 4.0                                        # raw (un-normalised) network outputs of varying norm
        x[1] = 0.0                                           # the epsilon clamp of l2_normalize
        x[2] *= 1e-9
        out = {"x": x, "labels": y.astype(np.int64)}
        o1, o2, tar = (rng.standard_normal((B, C)).astype(np.float32) * 2 for _ in range(3))
        o2[np.arange(B)[::2], y[::2]] += 7.0                 # confident + correct: mask = 1, relu(p - alpha) > 0
        out.update(le_out1=o1, le_out2=o2, le_tar=tar)
        for fx, (u, ll) in refs.items():
            s = "_" + fx[-2:]
            y_true = E[y]                                    # learn_image_embeddings.py:48-50 (cast to floatx on feed)
            xhat = u.l2norm(x)                               # utils.py:125-127
            out["xhat" + s] = xhat
            out["inv_correlation" + s] = u.inv_correlation(y_true.astype(fx), xhat)                       # utils.py:44-46
            out["squared_distance" + s] = u.squared_distance(y_true.astype(fx), x)                        # utils.py:34-36
            out["mean_distance" + s] = u.mean_distance(y_true.astype(fx), x)                              # utils.py:39-41
            for k in (1, 5):
                out["max_sim_acc%d" % k + s] = u.nn_accuracy(E, True, k)(y_true.astype(fx), xhat)         # utils.py:87-95
                out["nn_accuracy%d" % k + s] = u.nn_accuracy(E, False, k)(y_true.astype(fx), x)           # utils.py:73-85
            out["devise_ranking_loss" + s] = u.devise_ranking_loss(E, 0.1)(y_true.astype(fx), xhat)       # utils.py:103-122
            out["labelembed_loss" + s] = ll.labelembed_loss(o1, o2, tar, y.astype(fx), num_classes=C)     # learn_labelembedding.py:21-37
        np.savez_compressed(os.path.join(OUT, "loss_ref_%s.npz" % key), **out)
        print("loss_ref", key, x.shape, "acc", float(out["max_sim_acc1_32"].mean()), float(out["nn_accuracy1_32"].mean()))


def lr_schedule_goldens():
    """Learning-rate trajectories of the reference's schedules (utils.py:288-399, clr_callback.py, sgdr_callback.py)."""
    from oracle import keras_stub
    u = ref_import.import_reference("utils")
    out = {}

    class _Model(object):
        def __init__(self, lr):
            self.optimizer = type("Opt", (), {})()
            self.optimizer.lr = keras_stub.Variable(lr)

    cbs, n = u.get_lr_schedule("SGDR", 50000, 100, {"sgdr_base_len": 4, "sgdr_mul": 2, "sgdr_max_lr": 0.1})
    m = _Model(0.1)
    cbs[0].set_model(m)
    cbs[0].on_train_begin()
    lrs = []
    for ep in range(30):
        lrs.append(m.optimizer.lr.value)
        cbs[0].on_epoch_end(ep, {})
    out["sgdr_lr_per_epoch"], out["sgdr_epochs"] = np.array(lrs), n

    cbs, n = u.get_lr_schedule("CLR", 1000, 100, {"clr_step_len": 2, "clr_min_lr": 1e-5, "clr_max_lr": 0.1})
    m = _Model(0.1)
    cbs[0].set_model(m)
    cbs[0].on_train_begin()
    lrs = []
    for it in range(100):
        lrs.append(m.optimizer.lr.value)
        cbs[0].on_batch_end(it, {})
    out["clr_lr_per_batch"], out["clr_epochs"] = np.array(lrs), n

    cbs, n = u.get_lr_schedule("SGD", 50000, 100, {"sgd_schedule": "1:0.1,31:0.01,41:0.001,50"})
    out["sgd_schedule_lr"], out["sgd_schedule_epochs"] = np.array([cbs[0].schedule(ep, 0.5) for ep in range(50)]), n
    cbs, n = u.get_lr_schedule("ResNet-Schedule", 50000, 100, {})
    out["resnet_schedule_lr"], out["resnet_schedule_epochs"] = np.array([cbs[0].schedule(ep) for ep in range(164)]), n
    cbs, n = u.get_lr_schedule("SGD", 50000, 100, {})
    out["sgd_plateau_epochs"], out["sgd_plateau_patience"], out["sgd_plateau_min_lr"] = n, cbs[0].kw["patience"], cbs[0].kw["min_lr"]
    np.savez_compressed(os.path.join(OUT, "lr_schedules.npz"), **out)
    print("lr schedules", {k: np.shape(v) for k, v in out.items()})


def topk_goldens(er):
    """topk_head_*.npz: the HEAD (first 256 entries) of the imported reference's rankings on D > 448 problems large
    enough for top-251 lists (the sharded-gallery / clipped-AHP consumers; BASELINE configs[4] is D = 1000), with the probed
    BLAS K-block list.  evaluate_retrieval.py:57-67."""
    import hashlib
    from oracle.retrieval_oracle import topk_feature_matrix
    e_inet = np.load(os.path.join(OUT, "imagenet_mintree_unitsphere.npz"))["embedding"]
    n = 640
    for name, kind, seed, norm in (("d1000_cos", "inet", 11, True), ("d1000_euc", "inet", 12, False), ("d555_cos", "gauss", 13, True),
                                   ("d555_euc", "gauss", 14, False)):
        feat = topk_feature_matrix(kind, seed, n, e_inet)
        rank = ref_ranking(er, feat, norm)
        kb = probe_kblocks(feat)
        np.savez_compressed(os.path.join(OUT, "topk_head_%s.npz" % name), kind=np.array(kind), seed=np.int64(seed), n=np.int64(n),
                            sha1=np.array(hashlib.sha1(feat.tobytes()).hexdigest()), normalize=np.bool_(norm),
                            ref_head=rank[:, :256].astype(np.int16), kblocks=np.array(kb, dtype=np.int32))
        print("topk", name, feat.shape, norm, "kblocks", kb)


def cifar_pipeline_goldens():
    """cifar_pipeline.npz: the reference's OWN in-memory data path -- datasets/cifar.py:9-84 (pickle parsing, class restriction /
    re-enumeration, NHWC reshape) and datasets/common.py:635-796 (TinyDatasetGenerator: featurewise statistics of the training set,
    DataSequence batches, compose_batch = random_transform + standardize per image) -- imported unmodified and run on a small
    synthetic CIFAR-100 pickle pair.  Only keras.preprocessing.image.ImageDataGenerator is a stand-in (oracle/keras_stub.py).
    Stored: the pickles' contents, the statistics, every un-augmented test / train batch (all classes, and restricted to 7
    re-enumerated classes), and augmented training images together with the (row shift, column shift, flip) the generator drew,
    so that the device augmentation can be checked on the SAME parameters."""
    import tempfile
    ds = ref_import.import_reference_datasets()
    rng = np.random.default_rng(31)
    n_train, n_test = 240, 64
    raw_train = rng.integers(0, 256, size=(n_train, 3072), dtype=np.uint8)
    raw_test = rng.integers(0, 256, size=(n_test, 3072), dtype=np.uint8)
    # smooth images (augmentation with bilinear shifts is only meaningful on non-noise content): low-pass the noise
    def smooth(raw):
        img = raw.reshape(-1, 3, 32, 32).astype(np.float32)
        for _ in range(3):
            img = (img + np.roll(img, 1, 2) + np.roll(img, -1, 2) + np.roll(img, 1, 3) + np.roll(img, -1, 3)) / 5.0
        return np.clip(np.round((img - img.min()) / (img.max() - img.min()) * 255.0), 0, 255).astype(np.uint8).reshape(-1, 3072)
    raw_train, raw_test = smooth(raw_train), smooth(raw_test)
    y_train = rng.integers(0, 100, size=n_train).tolist()
    y_train[:100] = list(range(100))                     # every class present (the reference takes max(y_train) + 1 classes)
    y_test = rng.integers(0, 100, size=n_test).tolist()
    out = dict(raw_train=raw_train, raw_test=raw_test, y_train=np.array(y_train), y_test=np.array(y_test))
    with tempfile.TemporaryDirectory() as tmp:
        for name, raw, lab in (("train", raw_train, y_train), ("test", raw_test, y_test)):
            with open(os.path.join(tmp, name), "wb") as f:
                pickle.dump({b"data": raw, b"fine_labels": lab}, f)
        gen = ds.CifarGenerator(tmp)
        out["mean"], out["std"] = gen.image_generator.mean, gen.image_generator.std
        out["num_classes"], out["num_train"], out["num_test"] = gen.num_classes, gen.num_train, gen.num_test
        for split, seq in (("test", gen.test_sequence(batch_size=24)), ("train", gen.train_sequence(batch_size=50, shuffle=False, augment=False))):
            xs, ys = zip(*[seq[i] for i in range(len(seq))])
            X_all = np.concatenate(xs)
            keep = len(X_all) if split == "test" else 50         # the training split: first batch in full, per-image checksums for the rest
            out[split + "_X"], out[split + "_y"] = X_all[:keep], np.concatenate(ys)
            out[split + "_image_sums"] = X_all.astype(np.float64).sum(axis=(1, 2, 3))
            out[split + "_batches"] = len(seq)
        # batch_transform hook (learn_image_embeddings.transform_inputs' place)
        seq = gen.test_sequence(batch_size=24, batch_transform=lambda X, y, scale: (X * scale, y + 1), batch_transform_kwargs={"scale": 2.0})
        out["transformed_X0"], out["transformed_y0"] = seq[0]
        # augmented training images with the drawn parameters
        np.random.seed(5)
        aug, params = [], []
        for j in range(16):
            x = gen.image_generator.random_transform(gen.X_train[j].astype("float32"))
            params.append(gen.image_generator.last_transform)
            aug.append(gen.image_generator.standardize(x))
        out["aug_X"], out["aug_params"] = np.stack(aug), np.array(params, dtype=np.float64)
        # class restriction + re-enumeration (datasets/cifar.py:57-72)
        classes = [3, 17, 20, 42, 56, 77, 99]
        gen_r = ds.CifarGenerator(tmp, classes=classes, reenumerate=True)
        seq = gen_r.test_sequence(batch_size=1000)
        out["restricted_classes"] = np.array(classes)
        out["restricted_test_X"], out["restricted_test_y"] = seq[0] if len(seq) else (np.zeros((0, 32, 32, 3), np.float32), np.zeros((0,), np.int64))
        out["restricted_mean"], out["restricted_num_train"] = gen_r.image_generator.mean, gen_r.num_train
    np.savez_compressed(os.path.join(OUT, "cifar_pipeline.npz"), **out)
    print("cifar_pipeline", out["test_X"].shape, out["train_X"].shape, "mean", out["mean"].ravel(), "restricted", out["restricted_test_X"].shape)


def big_retrieval_golden(er):
    """Round 5: ONE larger reference-produced retrieval fixture -- 4,096 clustered items (CIFAR-100 class embedding + noise, 64 exact
    duplicates), D = 100, both branches of the reference (cosine and the CLI-default Euclidean) -- so that the reference-ranking gate
    is exercised beyond 256 rows.  The file keeps the features and the reference's ranking of 128 evenly spread + random query rows per
    branch (uint16 indices); the rankings are the output of the imported, unmodified evaluate_retrieval.pairwise_retrieval."""
    e_cifar, _ = load_embedding("cifar100.unitsphere")
    rng = np.random.default_rng(11)
    n = 4096
    y = rng.integers(0, 100, size=n)
    x = (e_cifar[y] + 0.1 * rng.standard_normal((n, 100))).astype(np.float32)
    x[n - 64:] = x[:64]                                   # exact duplicate rows: exact ties in every ranking
    rows = np.unique(np.concatenate([np.linspace(0, n - 1, 96).astype(np.int64), rng.integers(0, n, size=40), np.arange(n - 8, n)]))[:128]
    out = {"features": x, "labels": y.astype(np.int16), "rows": rows.astype(np.int32)}
    for name, norm in (("cos", True), ("euc", False)):
        rank = ref_ranking(er, x, norm)
        assert rank.shape == (n, n) and rank.max() < 65536
        out["ref_ranking_rows_" + name] = rank[rows].astype(np.uint16)
        print("big", name, rank.shape)
    np.savez_compressed(os.path.join(OUT, "bigretrieval_cluster.npz"), **out)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "big":       # only the round-5 large retrieval fixture
        big_retrieval_golden(ref_import.import_reference("evaluate_retrieval"))
        return
    if len(sys.argv) > 1 and sys.argv[1] == "cifar":     # only the round-3 data-pipeline fixture
        cifar_pipeline_goldens()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "topk":      # only the round-3 top-k fixtures (everything else untouched)
        topk_goldens(ref_import.import_reference("evaluate_retrieval"))
        return
    os.makedirs(OUT, exist_ok=True)
    er = ref_import.import_reference("evaluate_retrieval")
    ch = ref_import.import_reference("class_hierarchy")

    # ---------------------------------------------------------------- embeddings (data fixtures)
    emb = {}
    for name in ("cifar100.unitsphere", "cifar100.glove", "nab.sim8", "cub_balanced.unitsphere"):
        e, lab = load_embedding(name)
        key = name.replace(".", "_")
        emb[key] = e
        emb[key + "__ind2label"] = lab
    np.savez_compressed(os.path.join(OUT, "embeddings.npz"), **emb)
    e_cifar = emb["cifar100_unitsphere"]

    # ---------------------------------------------------------------- retrieval cases
    rng = np.random.default_rng(0)
    cases = {}
    x = rng.standard_normal((256, 100)).astype(np.float32)
    cases["gauss_cos"] = (x, True)
    cases["gauss_euc"] = (x, False)
    # "trained-like": class embedding + noise, plus exact duplicate rows (forces exact ties)
    y = rng.integers(0, 100, size=224)
    xc = (e_cifar[y] + 0.1 * rng.standard_normal((224, 100))).astype(np.float32)
    xc[200:224] = xc[0:24]
    cases["cluster_cos"] = (xc, True)
    cases["cluster_euc"] = (xc, False)
    cases["d7_cos"] = (rng.standard_normal((96, 7)).astype(np.float32), True)       # pairwise-sum n < 8 path
    cases["d130_euc"] = (rng.standard_normal((96, 130)).astype(np.float32), False)  # pairwise-sum recursion
    cases["d200_cos"] = (rng.standard_normal((128, 200)).astype(np.float32), True)
    for name, (feat, norm) in cases.items():
        rank = ref_ranking(er, feat, norm)
        np.savez_compressed(os.path.join(OUT, "retrieval_%s.npz" % name), features=feat,
                            normalize=np.bool_(norm), ref_ranking=rank.astype(np.int16))
        print(name, feat.shape, norm, rank.shape)

    # dict input with non-trivial ids (evaluate_retrieval.py:46-49)
    ids = rng.permutation(1000)[:64] + 5
    fd = {int(i): rng.standard_normal(32).astype(np.float32) for i in ids}
    rank = ref_ranking(er, {"feat": fd}["feat"], True)
    np.savez_compressed(os.path.join(OUT, "retrieval_dict_ids.npz"), ids=np.array(list(fd.keys())),
                        features=np.stack(list(fd.values())), normalize=np.bool_(True), ref_ranking=rank)

    # ---------------------------------------------------------------- hierarchy metrics
    with open(os.path.join(REF, "Cifar-Hierarchy", "cifar.parent-child.txt")) as f:
        edges = np.array([[int(t) for t in l.split()] for l in f if l.strip()], dtype=np.int32)
    hier = ch.ClassHierarchy.from_file(os.path.join(REF, "Cifar-Hierarchy", "cifar.parent-child.txt"),
                                       id_type=int)
    labels = y[:200].tolist()
    feats = xc[:200]
    ks = list(range(1, 51)) + [100]
    res = {}
    for norm in (True, False):
        gen = er.pairwise_retrieval(feats.copy(), normalize=norm, return_generator=True)
        for ahp in (True, 50):
            avg, _ = hier.hierarchical_precision(gen if ahp is True else
                                                 er.pairwise_retrieval(feats.copy(), normalize=norm),
                                                 labels, ks, compute_ahp=ahp, compute_ap=True,
                                                 all_ids=list(range(len(labels))))
            for m, v in avg.items():
                res["%s|norm=%d|ahp=%s" % (m, norm, ahp)] = v
    # pairwise class similarities (LUT the GPU metric kernels would use)
    wup = np.array([[hier.wup_similarity(a, b) for b in range(100)] for a in range(100)])
    lcs = np.array([[1.0 - hier.lcs_height(a, b) for b in range(100)] for a in range(100)])
    np.savez_compressed(os.path.join(OUT, "hierarchy_cifar.npz"), edges=edges, labels=np.array(labels),
                        features=feats, ks=np.array(ks), metric_names=np.array(list(res.keys())),
                        metric_values=np.array(list(res.values())), wup=wup, lcs=lcs,
                        heights=np.array([hier.heights[i] for i in sorted(hier.nodes)]),
                        nodes=np.array(sorted(hier.nodes)), max_height=hier.max_height)

    hierarchy_goldens_more(er, ch)

    # ---------------------------------------------------------------- loss golden (oracle; unpinned)
    rng = np.random.default_rng(1)
    xb = rng.standard_normal((128, 100)).astype(np.float32)
    yb = rng.integers(0, 100, size=128).astype(np.int64)
    fwd = loss_oracle.cosine_loss_fwd(xb, yb, e_cifar)
    w = np.full(128, 1.0 / 128)
    dx = loss_oracle.cosine_loss_bwd(xb, yb, e_cifar, w)
    acc = loss_oracle.nn_accuracy(e_cifar, True)(e_cifar[yb], fwd["xhat"])
    np.savez_compressed(os.path.join(OUT, "loss_cifar100.npz"), x=xb, labels=yb, loss_i=fwd["loss_i"],
                        loss=fwd["loss"], inv_norm=fwd["inv_norm"], dx=dx, acc=acc)

    # ---------------------------------------------------------------- imagenet_mintree.unitsphere (regenerated)
    e_inet = regenerate_imagenet_mintree()

    # ---------------------------------------------------------------- retrieval, D > 448 (BLAS K blocks)
    from oracle import retrieval_oracle as ro
    e_nab, _ = load_embedding("nab.unitsphere")
    rng = np.random.default_rng(5)
    big = {}
    yn = rng.integers(0, e_nab.shape[0], size=160)
    xn = (e_nab[yn] + 0.05 * rng.standard_normal((160, e_nab.shape[1]))).astype(np.float32)
    big["d555_cos"] = (xn, True)
    big["d555_euc"] = (xn, False)
    yi = rng.integers(0, 1000, size=112)
    xi = (e_inet[yi] + 0.03 * rng.standard_normal((112, 1000))).astype(np.float32)
    big["d1000_cos"] = (xi, True)
    big["d1000_euc"] = (xi, False)
    for name, (feat, norm) in big.items():
        rank = ref_ranking(er, feat, norm)
        kb = probe_kblocks(feat)
        np.savez_compressed(os.path.join(OUT, "retrieval_%s.npz" % name), features=feat, normalize=np.bool_(norm),
                            ref_ranking=rank.astype(np.int16), kblocks=np.array(kb, dtype=np.int32))
        print(name, feat.shape, norm, "kblocks", kb)

    # ---------------------------------------------------------------- loss / metric goldens from the reference's own lines
    emb["imagenet_mintree_unitsphere"] = e_inet.astype(np.float32)   # as committed (the fixture must be self-consistent)
    loss_reference_goldens(emb)
    lr_schedule_goldens()
    topk_goldens(er)
    cifar_pipeline_goldens()
    big_retrieval_golden(er)
    print("done ->", OUT)


if __name__ == "__main__":
    main()
