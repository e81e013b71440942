"""Drop-in for the reference's ``evaluate_retrieval.py`` (same CLI flags, same
``pairwise_retrieval`` signature and return convention), with the all-pairs distance + ranking
executed by the MI355X kernels of libsehip.so instead of NumPy/OpenBLAS/numexpr on the host.

reference: evaluate_retrieval.py:22-73 (pairwise_retrieval), :76-151 (reporting helpers),
:155-208 (CLI).  Differences a user can observe:

* ties in the ranking come back in canonical (distance, index) order -- the reference's
  ``np.argsort`` is unstable and returns them in arbitrary order;
* the ranking is produced query tile by query tile (the generator is genuinely lazy), so the
  N x N distance matrix never has to exist at once;
* there is no CPU fallback: without a ROCm device the call raises ``sehip.SehipError``;
* launched with one process per GPU (``python -m torch.distributed.run --nproc-per-node G evaluate_retrieval.py ...``)
  the queries are sharded across the ranks (gallery replicated, rows of a ranking are independent) and the per-query
  metrics are combined with one small all-reduce; rank 0 prints / writes / plots;
* ``--skip_ap`` (extension) drops the average-precision column; together with ``--clip_ahp`` no metric needs more than the
  head of each ranking, and under several ranks the SHARDED-GALLERY path is taken: per-shard fused distance + top-k,
  RCCL all-gather of the per-shard lists, k-way merge.
"""
import argparse
import os.path
import pickle
import warnings
from collections import OrderedDict

import numpy as np

try:
    from tqdm import tqdm
except ImportError:  # pragma: no cover
    def tqdm(it, **kwargs):
        return it


METRICS = ['P@1 (WUP)', 'P@10 (WUP)', 'P@50 (WUP)', 'P@100 (WUP)', 'AHP (WUP)',
           'P@1 (LCS_HEIGHT)', 'P@10 (LCS_HEIGHT)', 'P@50 (LCS_HEIGHT)', 'P@100 (LCS_HEIGHT)', 'AHP (LCS_HEIGHT)', 'AP']

# rows of the distance matrix ranked per launch group; bounds device memory to ~8 B x TILE x N
DEFAULT_TILE_BYTES = 4 << 30


def _as_feature_matrix(features):
    """Input normalisation of evaluate_retrieval.py:43-54: path | dict | {'feat': dict} | ndarray."""
    if isinstance(features, str):
        with open(features, 'rb') as feat_dump:
            features = pickle.load(feat_dump)
    if isinstance(features, dict):
        if 'feat' in features:
            features = features['feat']
        ind2id = np.array(list(features.keys()))
        features = np.stack(list(features.values()))
        if features.ndim > 2:
            raise ValueError('Feature matrix must be 2-dimensional. Actual shape: {}'.format(features.shape))
        owned = True
    else:
        ind2id = None
        owned = False
    return features, ind2id, owned


def host_blas_kblocks(d, q=448):
    """K-block list of OpenBLAS's level-3 drivers for a depth-``d`` product (GEMM_Q = ``q``; 448 for the Haswell / SkylakeX /
    Zen sgemm kernels): blocks of ``q`` while at least ``2 q`` remain, then the remainder split in two (first half rounded up)
    if it exceeds ``q``.  The reference's ``np.dot`` (evaluate_retrieval.py:59,62) restarts its fp32 FMA chain at every block,
    so for D > 448 bit-identical rankings need this list: ``pairwise_retrieval(..., kblocks=host_blas_kblocks(D))``.
    Verified against this image's NumPy/OpenBLAS for D up to 2048 (oracle/make_golden.py probes it for the fixtures)."""
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


def _resolve_kblocks(kblocks, d, warn=True):
    if kblocks is not None and not isinstance(kblocks, str) and len(kblocks) == 0:
        return None            # "already resolved to a single chain" (pairwise_retrieval -> ranking_tiles): no second warning
    if kblocks is None:
        if warn and d > 448:
            import warnings
            warnings.warn("D = {} > 448: the reference's np.dot (evaluate_retrieval.py:59,62) restarts its float32 FMA chain per BLAS "
                          "K block; with kblocks=None distances come from ONE chain and near-tie orders can differ from the "
                          "reference's.  Pass kblocks='openblas' (CLI: --kblocks openblas) for bit-identical rankings."
                          .format(d), stacklevel=3)
        return None
    if isinstance(kblocks, str):
        if kblocks.lower() != 'openblas':
            raise ValueError("kblocks must be None, 'openblas' or a list of block lengths summing to D")
        kblocks = host_blas_kblocks(d)
    kblocks = [int(v) for v in kblocks]
    if sum(kblocks) != d or min(kblocks) <= 0:
        raise ValueError('kblocks {} do not add up to the feature dimension {}'.format(kblocks, d))
    return kblocks if len(kblocks) > 1 else None


_tile_cache = {}   # device index -> {'pd': flat uint8 tensor, 'rk': flat uint8 tensor}: grow-only distance / rank buffers of ranking_tiles


def release_tile_cache(device=None):
    """Drop the cached distance / rank buffers of ``ranking_tiles`` (one device, default: all).  They are grow-only and live for the
    process: 2 x 10 GB after a 50,000-item all-pairs evaluation."""
    if device is None:
        _tile_cache.clear()
    else:
        import torch
        dev = torch.device(device)
        _tile_cache.pop(dev.index if dev.index is not None else torch.cuda.current_device(), None)


def _cached_rows(kind, rows, n, dtype, device):
    """``[rows, n]`` view (row pitch a multiple of 16 bytes, like ``sehip.empty_rows``) of the device's grow-only cached buffer
    ``kind``: a second evaluation in the same process -- the CLI loops over its --feat files -- pays no 10 GB allocation."""
    import torch
    esz = torch.empty((), dtype=dtype).element_size()
    per16 = 16 // esz
    pitch = (n + per16 - 1) // per16 * per16
    need = rows * pitch * esz
    key = device.index if device.index is not None else torch.cuda.current_device()
    slot = _tile_cache.setdefault(key, {})
    buf = slot.get(kind)
    if buf is None or buf.numel() < need:
        slot.pop(kind, None)
        buf = None                        # free the old buffer before the larger one is allocated
        buf = torch.empty((max(need, 16),), dtype=torch.uint8, device=device)
        slot[kind] = buf
    mat = buf[:need].view(dtype).view(rows, pitch)
    return mat if pitch == n else mat[:, :n]


def ranking_tiles(features, normalize=False, tile_rows=None, idx64=False, queries=None, kblocks=None, whole_if_fits=True,
                  prenormalized=False, idx16=False):
    """Generator over ``(first_row, rank_tile)`` with ``rank_tile`` an int32 (int64 if ``idx64``)
    DEVICE tensor ``[rows, N]``: the canonical ranking of queries ``first_row .. first_row+rows``.
    **A tile is valid until the next one is drawn**: distances and ranks live in grow-only per-device buffers that every tile and
    every later call reuses (``release_tile_cache`` frees them).

    ``features`` must already be a float32 device tensor ``[N, D]``; it is normalised in place when
    ``normalize`` is set (like the reference mutates its input, evaluate_retrieval.py:58).
    (``prenormalized``: the caller already did that -- normalising twice would change bits.)
    ``queries`` optionally restricts the query rows to ``range(*queries)``; ``kblocks`` (None | 'openblas' | list) makes
    the FMA chain restart per K block like the host BLAS the reference ran on (see ``host_blas_kblocks``).
    ``idx16``: uint16 ranks as an int16 tensor (galleries of at most 53,248 items; what ``hierarchical_precision_device`` asks
    for: half the bytes between the ranking and the metric kernel).
    ``whole_if_fits`` (default since round 6): rank all queries as ONE tile when distances + ranks (8 N^2 bytes) fit into a third of
    the device memory that is free or already held by the tile cache -- all-pairs then takes the symmetric distance kernel
    (3.2 instead of 5.4 ms at 50k x 50k), i.e. the kernels bench.py times.  The first call on a device pays for the two
    allocations once (0.24 s for 2 x 10 GB); with the cache later evaluations do not."""
    import torch
    import sehip

    n, _ = features.shape
    kblocks = _resolve_kblocks(kblocks, features.shape[1])
    if normalize:
        if not prenormalized:
            sehip.normalize_rows_(features)
        metric, sq = sehip.METRIC_COSINE, None
    else:
        metric, sq = sehip.METRIC_EUCLID, sehip.row_sqnorm(features)
    q0, q1 = (0, n) if queries is None else queries
    if tile_rows is None:
        tile_rows = max(128, min(n, (DEFAULT_TILE_BYTES // (8 * max(n, 1))) // 128 * 128))
        if whole_if_fits and features.is_cuda:
            # distances + ranks of all queries, plus what the ranking will ask of the (grow-only) workspace cache on top of what that
            # cache already holds -- up to ~3 GB for rows above 53,248 columns
            extra_ws = max(0, sehip.rank_rows_workspace_bytes(q1 - q0, n) - sehip.workspace_bytes(features.device))
            key = features.device.index if features.device.index is not None else torch.cuda.current_device()
            held = sum(int(b.numel()) for b in _tile_cache.get(key, {}).values())
            need = (4 + (2 if idx16 else (8 if idx64 else 4))) * n * (q1 - q0)
            if need + extra_ws <= (torch.cuda.mem_get_info(features.device)[0] + held) // 3:
                tile_rows = max(tile_rows, q1 - q0)
    rows_max = min(tile_rows, max(q1 - q0, 1))
    if features.is_cuda:
        pd = _cached_rows('pd', rows_max, n, torch.float32, features.device)
        rk = _cached_rows('rk', rows_max, n, torch.int16 if idx16 else (torch.int64 if idx64 else torch.int32), features.device)
    else:   # (CPU stand-ins of the tests never get here: the kernels need a device)
        pd = sehip.empty_rows(rows_max, n, torch.float32, features.device)
        rk = None
    for r0 in range(q0, q1, tile_rows):
        rows = min(tile_rows, q1 - r0)
        sehip.pairwise_dist(features[r0:r0 + rows], features, metric=metric,
                            sqa=None if sq is None else sq[r0:r0 + rows], sqb=sq, kblocks=kblocks, out=pd[:rows])
        yield r0, sehip.rank_rows(pd[:rows], idx64=idx64, idx16=idx16, out=None if rk is None else rk[:rows])


def pairwise_retrieval(features, normalize=False, return_generator=True, kblocks=None):
    """ Uses each image as query and retrieves its nearest neighbors.

    # Arguments (identical to the reference, evaluate_retrieval.py:22-41):

    - features: 2-d numpy array | dict id -> feature vector | dict with key 'feat' | path to a pickle of those.
    - normalize: Whether to L2-normalize the features.
    - return_generator: If True, a generator will be returned instead of a dictionary.
    - kblocks (extension): None | 'openblas' | list -- K-block list of the host BLAS to reproduce for D > 448.

    # Returns:
        generator (or dict) of ``(image ID, list of all image IDs ordered by increasing distance)``.
    """
    import torch
    import sehip  # raises SehipError if the HIP library is missing -- no CPU fallback

    features, ind2id, owned = _as_feature_matrix(features)
    if isinstance(features, np.ndarray) and features.dtype == np.float64:
        # the reference computes in the caller's dtype (evaluate_retrieval.py:57-67): float64 features give float64 distances and
        # their ranking.  The kernels are float32 (the dtype of every feature dump the reference writes,
        # learn_image_embeddings.py:270-275): say so instead of casting silently.
        warnings.warn("pairwise_retrieval: float64 features are ranked in float32 here (the reference would compute float64 distances); "
                      "orders can differ where float32 rounding creates or breaks near-ties.  Feature dumps of the reference are float32.",
                      RuntimeWarning, stacklevel=2)
    feats_h = np.ascontiguousarray(features, dtype=np.float32)
    sehip._lib.require_gpu()
    dev = torch.from_numpy(feats_h).cuda()
    kblocks = _resolve_kblocks(kblocks, feats_h.shape[1])
    if normalize:
        # like the reference (`features /= np.linalg.norm(...)`, evaluate_retrieval.py:58) the caller's array is normalised
        # in place at CALL time, before the first ranking is drawn from the generator
        sehip.normalize_rows_(dev)
        if not owned and isinstance(features, np.ndarray) and features.dtype == np.float32:
            np.copyto(features, dev.cpu().numpy())

    def gen():
        for r0, tile in ranking_tiles(dev, normalize, kblocks=kblocks or (), prenormalized=True):
            ranks = tile.cpu().numpy()
            for i in range(ranks.shape[0]):
                ret = ranks[i]
                if ind2id is not None:
                    yield ind2id[r0 + i], ind2id[ret].tolist()
                else:
                    yield r0 + i, ret.tolist()

    g = gen()
    return g if return_generator else dict(g)


def print_performance(perf, metrics=METRICS):
    """Console table: one row per feature file, one column per metric (4 decimals)."""
    name_w = max(map(len, perf))
    col_w = [max(len(m), 6) for m in metrics]
    header = [' ' * name_w] + [m.center(6) for m in metrics]
    print('\n' + ' | '.join(header))
    print('-' * (name_w + sum(3 + w for w in col_w)))
    for name, res in perf.items():
        cells = ['%*.4f' % (w, res[m]) for m, w in zip(metrics, col_w)]
        print(' | '.join([name.ljust(name_w)] + cells))
    print()


def write_performance(perf, csv_file, prec_type='LCS_HEIGHT'):
    """Semicolon-separated P@k table, k = 1.. as long as every result has that cut-off."""
    names = list(perf)
    lines = ['k;' + ';'.join(names)]
    k = 1
    while all('P@%d (%s)' % (k, prec_type) in perf[n] for n in names):
        lines.append(';'.join([str(k)] + [str(perf[n]['P@%d (%s)' % (k, prec_type)]) for n in names]))
        k += 1
    with open(csv_file, 'w') as f:
        f.write('\n'.join(lines) + '\n')


def plot_performance(perf, kmax=100, prec_type='LCS_HEIGHT', clip_ahp=None):
    """Figure 1: hierarchical precision@k curves; figure 2: (clipped) mAHP as horizontal bars."""
    import matplotlib.pyplot as plt

    ks = np.arange(1, kmax + 1)
    fig1, ax = plt.subplots()
    lowest = 1.0
    for name, res in perf.items():
        curve = np.array([res['P@%d (%s)' % (k, prec_type)] for k in ks])
        ax.plot(ks, curve, label=name)
        lowest = min(lowest, float(curve.min()))
    floor = np.floor(lowest * 20) / 20
    ax.set(xlabel='k', ylabel='Hierarchical Precision', xlim=(0, kmax), ylim=(floor if floor >= 0.3 else 0, 1))
    ax.grid(True)
    ax.legend(fontsize='x-small')

    key = 'AHP%s (%s)' % ('@%d' % clip_ahp if clip_ahp else '', prec_type)
    fig2, ax = plt.subplots()
    for pos, (name, res) in enumerate(perf.items()):
        ax.barh(pos + 0.5, res[key], 0.8)
        ax.text(0.01, pos + 0.5, name, va='center', ha='left', color='white', fontsize='small')
        ax.text(res[key] - 0.01, pos + 0.5, '{:.1%}'.format(res[key]), va='center', ha='right', color='white')
    ax.set(xlabel='Mean Average Hierarchical Precision', yticks=[])
    ax.grid(axis='x')
    plt.show()


def str2bool(v):
    """argparse type for the --norm flag: yes/true/t/y/1 and no/false/f/n/0 (case-insensitive)."""
    word = v.lower()
    if word in ('yes', 'true', 't', 'y', '1'):
        return True
    if word in ('no', 'false', 'f', 'n', '0'):
        return False
    raise argparse.ArgumentTypeError('Boolean value expected.')


def build_parser():
    """Same flags, defaults and grouping as the reference CLI (evaluate_retrieval.py:157-174)."""
    p = argparse.ArgumentParser(description='Hierarchical-precision evaluation of nearest-neighbour image retrieval (MI355X kernels).',
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    g = p.add_argument_group('Dataset')
    g.add_argument('--dataset', type=str, required=True, help='Dataset name (see datasets.get_data_generator).')
    g.add_argument('--data_root', type=str, required=True, help='Dataset root directory.')
    g.add_argument('--hierarchy', type=str, required=True, help='Text file with one "<parent> <child>" pair per line.')
    g.add_argument('--is_a', action='store_true', default=False, help='Lines of --hierarchy are "<child> <parent>" instead.')
    g.add_argument('--str_ids', action='store_true', default=False, help='Class IDs are strings (default: integers).')
    g.add_argument('--classes_from', type=str, default=None, help='Pickle with an "ind2label" item restricting/ordering the classes.')
    g = p.add_argument_group('Features')
    g.add_argument('--feat', type=str, action='append', required=True, help='Feature pickle {"feat": {image id: vector}}; repeatable.')
    g.add_argument('--label', type=str, action='append', help='Display name for the matching --feat.')
    g.add_argument('--norm', type=str2bool, action='append', help='L2-normalise the matching --feat (cosine ranking); default no.')
    g = p.add_argument_group('Output')
    g.add_argument('--plot_max', type=int, default=250, help='Largest k of the precision curve; 0 disables plotting.')
    g.add_argument('--prec_type', type=str, default='LCS_HEIGHT', choices=['WUP', 'LCS_HEIGHT'], help='Class-similarity measure for the curve/CSV.')
    g.add_argument('--clip_ahp', type=int, default=None, help='Compute AHP on the first CLIP_AHP ranks only.')
    g.add_argument('--csv', type=str, default=None, help='Write the P@k table to this CSV file.')
    g = p.add_argument_group('Extensions of this build (not in the reference)')
    g.add_argument('--skip_ap', action='store_true', default=False,
                   help='Do not compute AP; with --clip_ahp only the head of each ranking is needed (sharded-gallery top-k under several ranks).')
    g.add_argument('--kblocks', type=str, default=None, help="'openblas': restart the fp32 dot-product chain per OpenBLAS K block (D > 448).")
    return p


def init_distributed():
    """One process per GPU when launched by torch.distributed.run (WORLD_SIZE > 1): RCCL ('nccl') on ROCm devices, gloo
    otherwise (CPU tests).  Returns (rank, world)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if torch.cuda.is_available():
            torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')) % max(torch.cuda.device_count(), 1))
            dist.init_process_group('nccl')
        else:
            dist.init_process_group('gloo')
    return dist.get_rank(), dist.get_world_size()


def main(argv=None):
    from datasets import get_data_generator
    from class_hierarchy import ClassHierarchy

    args = build_parser().parse_args(argv)
    rank, world = init_distributed()

    if args.classes_from:
        with open(args.classes_from, 'rb') as f:
            embed_labels = pickle.load(f)['ind2label']
    else:
        embed_labels = None
    data_generator = get_data_generator(args.dataset, args.data_root, classes=embed_labels)
    labels_test = [embed_labels[lbl] for lbl in data_generator.labels_test] if embed_labels is not None else data_generator.labels_test

    id_type = str if args.str_ids else int
    hierarchy = ClassHierarchy.from_file(args.hierarchy, is_a_relations=args.is_a, id_type=id_type)

    ks = list(range(1, args.plot_max + 1))
    for k in [1, 10, 50, 100]:
        if (len(ks) == 0) or (ks[-1] < k):
            ks.append(k)
    perf = OrderedDict()
    for i, feat_dump in tqdm(enumerate(args.feat), total=len(args.feat), disable=rank != 0):
        feat_name = args.label[i] if (args.label is not None) and (i < len(args.label)) else os.path.splitext(os.path.basename(feat_dump))[0]
        normalize = args.norm[i] if (args.norm is not None) and (i < len(args.norm)) else False
        # reference: hierarchy.hierarchical_precision(pairwise_retrieval(feat_dump, normalize), labels_test, ks, ...)
        # (evaluate_retrieval.py:197-201).  Here rankings and metrics stay on the GPU: no N x N Python lists.
        features, ind2id, _ = _as_feature_matrix(feat_dump)
        # Several ranks: queries sharded (full rankings), or -- when no metric needs more than the head of a ranking
        # (--skip_ap with --clip_ahp) -- the gallery sharded with an all-gather + merge of per-shard top-k lists.
        perf[feat_name] = hierarchy.hierarchical_precision_device(
            features, labels_test, ks, compute_ahp=args.clip_ahp if args.clip_ahp else True, compute_ap=not args.skip_ap,
            normalize=normalize, ids=None if ind2id is None else ind2id.tolist(), distributed=world > 1,
            kblocks=args.kblocks, per_query=False)[0]      # the tables / plots below use the means only
    if rank != 0:
        return perf

    metrics = list(METRICS)
    if args.skip_ap:
        metrics.remove('AP')
    if args.clip_ahp:
        metrics[4] = 'AHP@{} (WUP)'.format(args.clip_ahp)
        metrics[9] = 'AHP@{} (LCS_HEIGHT)'.format(args.clip_ahp)
    print_performance(perf, metrics)
    if args.csv:
        write_performance(perf, args.csv, args.prec_type)
    if args.plot_max > 0:
        plot_performance(perf, args.plot_max, args.prec_type, args.clip_ahp)
    return perf


if __name__ == '__main__':
    main()
