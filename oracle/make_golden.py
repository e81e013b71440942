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
* loss_cifar100.npz   seeded inputs + loss oracle outputs (NOT from the reference: Keras/TF are
                      absent -- "parity unpinned", see oracle/loss_oracle.py)

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


def main():
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
    print("done ->", OUT)


if __name__ == "__main__":
    main()
