"""Class taxonomy + hierarchical retrieval metrics: the consumer of the rankings produced by
``evaluate_retrieval.pairwise_retrieval``.

Same public surface as the reference's ``class_hierarchy.ClassHierarchy``
(reference: class_hierarchy.py:7-367 -- ``from_file``/``save``, ``lcs``, ``wup_similarity``,
``lcs_height``, ``depth``, ``heights``/``max_height``, ``hierarchical_precision`` ...), but built
differently: node properties are computed once by memoised graph walks, and
``hierarchical_precision`` works on NumPy look-up tables (class x class similarity matrices gathered
along the ranking + prefix sums) instead of per-element Python dictionary look-ups.  The metric
definitions (Deng et al., CVPR 2011 hierarchical precision; area under the HP@k curve; average
precision) and every corner of the reference's bookkeeping are kept, so the numbers agree with the
reference to float64 round-off (tests/test_class_hierarchy.py checks against values produced by the
imported reference).
"""
import types

import os

import numpy as np

_trapz = getattr(np, "trapezoid", None) or np.trapz


class ClassHierarchy(object):
    """A DAG of classes given as parent->children / child->parents adjacency dictionaries."""

    def __init__(self, parents, children):
        self.parents = parents
        self.children = children
        self.nodes = set(parents) | set(children)
        self._depth = {False: {}, True: {}}
        self._anc_depth = {False: {}, True: {}}
        self._anc_dist = {}
        self._lcs_cache = {}
        self._wup_cache = {}
        self._luts = {}
        self.heights = {}
        for node in self.nodes:
            self._height(node)
        self.max_height = max(self.heights.values())

    # ------------------------------------------------------------------ construction / IO

    @classmethod
    def from_file(cls, rel_file, is_a_relations=False, id_type=str):
        """Reads "<parent> <child>" lines ("<child> <parent>" with ``is_a_relations``)."""
        parents, children = {}, {}
        with open(rel_file) as f:
            for line in f:
                line = line.strip()
                if not line:
                    continue
                first, second = (id_type(tok) for tok in line.split(maxsplit=1))
                parent, child = (second, first) if is_a_relations else (first, second)
                parents.setdefault(child, []).append(parent)
                children.setdefault(parent, []).append(child)
        return cls(parents, children)

    def save(self, filename, is_a_relations=False):
        """Writes the edges back in the format ``from_file`` reads."""
        with open(filename, 'w') as f:
            if is_a_relations:
                f.writelines('{} {}\n'.format(c, p) for c, ps in self.parents.items() for p in ps)
            else:
                f.writelines('{} {}\n'.format(p, c) for p, cs in self.children.items() for c in cs)

    # ------------------------------------------------------------------ node properties

    def _height(self, node):
        """Longest downward path to a leaf (leaves: 0)."""
        h = self.heights.get(node)
        if h is None:
            kids = self.children.get(node, ())
            h = 1 + max((self._height(k) for k in kids), default=-1) if node in self.children else 0
            self.heights[node] = h
        return h

    def is_tree(self):
        return all(len(ps) <= 1 for ps in self.parents.values())

    def depth(self, id, use_min_depth=False):
        """Roots have depth 1; otherwise 1 + max (or min) over the parents' depths."""
        memo = self._depth[use_min_depth]
        if id not in memo:
            ps = self.parents.get(id) or []
            if not ps:
                memo[id] = 1
            else:
                pick = min if use_min_depth else max
                memo[id] = 1 + pick(self.depth(p, use_min_depth) for p in ps)
        return memo[id]

    def all_hypernym_depths(self, id, use_min_depth=False):
        """{ancestor (incl. id): depth of that ancestor}."""
        memo = self._anc_depth[use_min_depth]
        if id not in memo:
            out = {}
            for p in self.parents.get(id) or []:
                out.update(self.all_hypernym_depths(p, use_min_depth))
            out[id] = self.depth(id, use_min_depth)
            memo[id] = out
        return memo[id]

    def all_hypernym_distances(self, id):
        """{ancestor (incl. id): fewest upward edges from id}."""
        if id not in self._anc_dist:
            out = {id: 0}
            for p in self.parents.get(id, ()):
                for anc, dist in self.all_hypernym_distances(p).items():
                    if dist + 1 < out.get(anc, float('inf')):
                        out[anc] = dist + 1
            self._anc_dist[id] = out
        return self._anc_dist[id]

    def root_paths(self, id):
        """Every path from a direct parent of ``id`` up to a root."""
        paths = []
        for p in self.parents.get(id, ()):
            above = self.root_paths(p)
            paths.extend([[p] + tail for tail in above] if above else [[p]])
        return paths

    # ------------------------------------------------------------------ pairwise class relations

    def lcs(self, a, b, use_min_depth=False):
        """Deepest common ancestor (``None`` if there is none).  Among equally deep candidates --
        only possible in non-tree hierarchies, where the reference's pick is arbitrary -- the one
        with the smallest height, then the smallest repr, is chosen deterministically."""
        key = (a, b)
        if key not in self._lcs_cache:
            da = self.all_hypernym_depths(a, use_min_depth)
            common = da.keys() & self.all_hypernym_depths(b, use_min_depth).keys()
            best = None
            if common:
                top = max(da[h] for h in common)
                best = min((h for h in common if da[h] == top), key=lambda h: (self.heights[h], repr(h)))
            self._lcs_cache[(a, b)] = self._lcs_cache[(b, a)] = best
        return self._lcs_cache[key]

    def shortest_path_length(self, a, b):
        da, db = self.all_hypernym_distances(a), self.all_hypernym_distances(b)
        return min((da[h] + db[h] for h in da.keys() & db.keys()), default=None)

    def wup_similarity(self, a, b):
        """Wu-Palmer: 2 depth(lcs) / (depth_via_lcs(a) + depth_via_lcs(b))."""
        key = (a, b)
        if key not in self._wup_cache:
            anc = self.lcs(a, b)
            ds = self.depth(anc)
            d1 = ds + self.shortest_path_length(a, anc)
            d2 = ds + self.shortest_path_length(b, anc)
            self._wup_cache[(a, b)] = self._wup_cache[(b, a)] = (2.0 * ds) / (d1 + d2)
        return self._wup_cache[key]

    def lcs_height(self, a, b):
        """height(lcs(a, b)) / height of the hierarchy (a dissimilarity in [0, 1])."""
        return self.heights[self.lcs(a, b)] / self.max_height

    def similarity_tables(self, classes):
        """(WUP, 1 - LCS height) as float64 [C, C] matrices over ``classes`` (cached)."""
        key = tuple(classes)
        if key not in self._luts:
            c = len(key)
            wup = np.empty((c, c))
            lcs = np.empty((c, c))
            for i, a in enumerate(key):
                for j in range(i, c):
                    b = key[j]
                    wup[i, j] = wup[j, i] = self.wup_similarity(a, b)
                    lcs[i, j] = lcs[j, i] = 1.0 - np.array(self.heights[self.lcs(a, b)]) / self.max_height
            self._luts[key] = (wup, lcs)
        return self._luts[key]

    # ------------------------------------------------------------------ retrieval metrics

    def hierarchical_precision(self, retrieved, labels, ks=[1, 10, 50, 100], compute_ahp=False, compute_ap=False,
                               ignore_qids=True, all_ids=None):
        """Hierarchical precision@k, area under that curve (AHP / AHP@K) and AP per query + means.

        Arguments and return value as in the reference (class_hierarchy.py:211-316):
        ``retrieved`` maps query id -> ranked id list (dict or generator of pairs), ``labels`` maps
        image id -> class label; returns ``(means, per_query)`` with metric names ``"P@K (WUP)"``,
        ``"P@K (LCS_HEIGHT)"``, ``"AHP[@K] (...)"`` and ``"AP"``.

        Bookkeeping kept from the reference: the best-possible cumulative similarity of a query
        class is taken from the FIRST query of that class (full list, query included); removing the
        query shifts that curve left at the query's rank and subtracts its self-similarity of 1;
        ids missing from a ranking are appended in ``all_ids`` order."""
        ks = [ks] if isinstance(ks, int) else list(ks)
        kmax = max(ks)
        ahp_clip = None if isinstance(compute_ahp, bool) else int(compute_ahp)
        if ahp_clip is not None:
            kmax = max(kmax, ahp_clip)
        full_lists = compute_ahp is True

        names = ['P@{} ({})'.format(k, t) for k in ks for t in ('WUP', 'LCS_HEIGHT')]
        ahp_names = ()
        if compute_ahp:
            sfx = '' if ahp_clip is None else '@{}'.format(ahp_clip)
            ahp_names = ('AHP{} (WUP)'.format(sfx), 'AHP{} (LCS_HEIGHT)'.format(sfx))
        prec = {n: {} for n in names}
        prec.update({n: {} for n in ahp_names})
        if compute_ap:
            prec['AP'] = {}

        # label -> row/column index of the similarity tables
        get_label = labels.__getitem__
        label_of = {}
        best_cum = {}
        class_list, class_pos = [], {}

        def cls_index(lbl):
            if lbl not in class_pos:
                class_pos[lbl] = len(class_list)
                class_list.append(lbl)
            return class_pos[lbl]

        lut_cache = {'n': -1, 'wup': None, 'lcs': None}

        def tables():
            if lut_cache['n'] != len(class_list):
                lut_cache['wup'], lut_cache['lcs'] = self.similarity_tables(class_list)
                lut_cache['n'] = len(class_list)
            return lut_cache['wup'], lut_cache['lcs']

        items = retrieved if isinstance(retrieved, types.GeneratorType) else retrieved.items()
        for qid, ret in items:
            lbl = get_label(qid)
            if all_ids and (len(ret) < len(all_ids)):
                seen = set(ret)
                ret = list(ret) + [i for i in all_ids if i not in seen]
            need_full = full_lists or (lbl not in best_cum)
            head = ret if need_full else ret[:kmax + 1]
            cols = np.fromiter((cls_index(label_of.setdefault(r, get_label(r)) if r in label_of else
                                          label_of.setdefault(r, get_label(r))) for r in head), dtype=np.int64, count=len(head))
            qi = cls_index(lbl)
            wup_t, lcs_t = tables()
            wup = wup_t[qi, cols]
            lcs = lcs_t[qi, cols]
            if lbl not in best_cum:
                best_cum[lbl] = (np.cumsum(np.sort(wup)[::-1]), np.cumsum(np.sort(lcs)[::-1]))
            cum_best_wup, cum_best_lcs = best_cum[lbl]

            q_pos = None
            if ignore_qids:
                try:
                    q_pos = ret.index(qid)
                except ValueError:
                    q_pos = None
                if q_pos is not None and q_pos < len(wup):
                    wup = np.delete(wup, q_pos)
                    lcs = np.delete(lcs, q_pos)
                    cum_best_wup = np.concatenate((cum_best_wup[:q_pos], cum_best_wup[q_pos + 1:] - 1.0))
                    cum_best_lcs = np.concatenate((cum_best_lcs[:q_pos], cum_best_lcs[q_pos + 1:] - 1.0))

            cw, cl = np.cumsum(wup), np.cumsum(lcs)
            for k in ks:
                kk = min(k, len(cw))
                prec['P@{} (WUP)'.format(k)][qid] = (cw[kk - 1] if kk else 0.0) / cum_best_wup[k - 1]
                prec['P@{} (LCS_HEIGHT)'.format(k)][qid] = (cl[kk - 1] if kk else 0.0) / cum_best_lcs[k - 1]
            if compute_ahp:
                if ahp_clip is None:
                    prec[ahp_names[0]][qid] = _trapz(cw / cum_best_wup, dx=1. / len(wup))
                    prec[ahp_names[1]][qid] = _trapz(cl / cum_best_lcs, dx=1. / len(lcs))
                else:
                    prec[ahp_names[0]][qid] = _trapz(cw[:ahp_clip] / cum_best_wup[:ahp_clip], dx=1. / ahp_clip)
                    prec[ahp_names[1]][qid] = _trapz(cl[:ahp_clip] / cum_best_lcs[:ahp_clip], dx=1. / ahp_clip)
            if compute_ap:
                rel = np.fromiter((label_of.setdefault(r, get_label(r)) == lbl for r in ret), dtype=bool, count=len(ret))
                if ignore_qids and q_pos is not None:
                    rel = np.delete(rel, q_pos)
                prec['AP'][qid] = _average_precision(rel)

        return {metric: sum(values.values()) / len(values) for metric, values in prec.items()}, prec


    def hierarchical_precision_device(self, features, labels, ks=[1, 10, 50, 100], compute_ahp=False, compute_ap=False,
                                      normalize=False, ids=None, tile_rows=None, distributed=False, group=None, kblocks=None,
                                      gather_per_query=True, kernels=None, per_query=True, head_via_topk=True):
        """``hierarchical_precision(pairwise_retrieval(features, normalize), labels, ...)`` (ignore_qids = True, every
        image is query and gallery item) without leaving the GPU: the rankings stay device tensors
        (``evaluate_retrieval.ranking_tiles``) and the per-query gather + prefix sums run in
        ``se_hierarchical_precision`` instead of the reference's Python loop (class_hierarchy.py:257-314, ~0.6 h at
        N = 50k) -- and the 330 s ``.tolist()`` hand-off of evaluate_retrieval.py:69-73 disappears.

        ``features``: float32 ``[N, D]`` array or (device) tensor (a device copy is normalised when ``normalize``);
        ``labels``: class label of image ``ids[i]`` (``ids`` defaults to ``range(N)``), as a sequence or a mapping.
        Returns ``(means, per_query)`` exactly like ``hierarchical_precision``; ``per_query=False`` returns ``(means, None)`` with
        the means taken on the device (the CLI only prints means: at N = 50k and 250 cut-offs the per-query dictionaries are
        25 million Python floats, seconds of host time after milliseconds of kernels).

        ``distributed`` (one process per GPU, ``torch.distributed`` initialised): every rank holds all features.
        * metrics that need full rankings (AP, un-clipped AHP): the QUERIES are sharded -- rank r ranks rows
          ``shard_bounds(N, G)[r]`` against the replicated gallery, no data-path collective; the per-query metric rows
          are all-gathered (``gather_per_query``) or only their sums all-reduced;
        * otherwise (P@k and AHP@clip only) the GALLERY is sharded: per-shard fused distance + top-L with
          L = max(ks, clip) + 1, RCCL all-gather of the ``(distance, global index)`` lists, canonical k-way merge
          (``sharded_retrieval.sharded_topk``), then each rank scores its share of the queries.
        With one process the same fused top-L path serves P@k / AHP@clip-only requests (``head_via_topk``; the full-ranking
        path otherwise).  ``kblocks`` (None | 'openblas' | list: the BLAS K-block list for D > 448) reaches BOTH paths -- the
        top-L kernels restart their FMA chain per block exactly like the full-ranking ones, so one process and G processes
        return the same near-tie orders.
        ``kernels`` (tests): CPU stand-ins ``{'ranking_tiles', 'hierarchical_precision', 'local_topk', 'merge', 'device'}``."""
        import torch
        from sharded_retrieval import shard_bounds, sharded_topk
        kernels = dict(kernels or {})
        native_metrics = 'hierarchical_precision' not in kernels
        native_ranking = 'ranking_tiles' not in kernels
        if 'ranking_tiles' not in kernels or 'hierarchical_precision' not in kernels:
            import sehip
            from evaluate_retrieval import ranking_tiles
            kernels.setdefault('ranking_tiles', ranking_tiles)
            kernels.setdefault('hierarchical_precision', sehip.hierarchical_precision)

        ks = [ks] if isinstance(ks, int) else list(ks)
        ahp_clip = None if isinstance(compute_ahp, bool) else int(compute_ahp)
        n = int(features.shape[0])
        ids = list(range(n)) if ids is None else list(ids)
        lab = [labels[i] for i in ids]
        class_list = sorted(set(lab), key=lambda c: (str(type(c)), c))
        pos = {c: i for i, c in enumerate(class_list)}
        cls_h = np.array([pos[c] for c in lab], dtype=np.int32)
        wup_t, lcs_t = self.similarity_tables(class_list)
        # best-possible cumulative similarity per query class: descending-sorted similarities of the whole gallery.  The C x N float64
        # curves are built ON THE DEVICE (round 6: 180 host cumsums of 50,000 entries + 80 MB of host-to-device copies were 25 of the 56 ms
        # of a 50,000-item evaluation): per class the C similarity values in descending order, each repeated by its class count,
        # then one float64 prefix sum per row.
        counts = np.bincount(cls_h, minlength=len(class_list))
        dev = kernels.get('device') or torch.device('cuda', torch.cuda.current_device())

        def best_curves(table):
            table = np.asarray(table, dtype=np.float64)
            order = np.argsort(-table, axis=1, kind='stable')
            vals = torch.from_numpy(np.take_along_axis(table, order, axis=1)).to(dev)
            reps = torch.from_numpy(counts[order].astype(np.int64)).to(dev)
            flat = torch.repeat_interleave(vals.reshape(-1), reps.reshape(-1), output_size=len(class_list) * n)
            return flat.view(len(class_list), n).cumsum(dim=1)

        best_w, best_l = best_curves(wup_t), best_curves(lcs_t)

        import torch.distributed as dist
        world = dist.get_world_size(group) if (distributed and dist.is_initialized()) else 1
        rank = dist.get_rank(group) if world > 1 else 0
        if torch.is_tensor(features):    # features straight from the network (learn_image_embeddings feature extraction): stay on the device
            feats = features.detach().to(device=dev, dtype=torch.float32).contiguous().clone()
        else:
            feats = torch.from_numpy(np.ascontiguousarray(features, dtype=np.float32)).to(dev)
        cls_d = torch.from_numpy(cls_h).to(dev)
        qidx_d = torch.arange(n, dtype=torch.int32, device=dev)
        ks_d = torch.tensor(ks, dtype=torch.int32, device=dev)
        ahp_len = -1 if not compute_ahp else (0 if ahp_clip is None else ahp_clip)
        head_only = (not compute_ap) and (not compute_ahp or ahp_clip is not None)
        q0, q1 = shard_bounds(n, world)[rank] if world > 1 else (0, n)
        ncol = 2 * len(ks) + 3
        outs = []

        def curves(args_d):     # the best curves pre-divided for se_hierarchical_precision (the CPU stand-ins of the tests divide themselves)
            if not native_metrics:
                return {}
            import sehip
            return {'curves': sehip.hprec_reciprocal_curves(args_d[2], args_d[3])}

        if head_only and (world > 1 or head_via_topk):
            # ---- top-L lists are enough for every requested metric: fused distance + top-L (the N x N matrix is never written),
            #      over this rank's shard of the gallery when there are several ranks ----
            L = min(n, max(ks + [ahp_clip or 0]) + 1)
            args_d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (wup_t, lcs_t)] + [best_w[:, :L + 1].contiguous(), best_l[:, :L + 1].contiguous()]
            g0, g1 = shard_bounds(n, world)[rank]
            if 'local_topk' not in kernels:
                import sehip
                if normalize:
                    sehip.normalize_rows_(feats)
                metric = sehip.METRIC_COSINE if normalize else sehip.METRIC_EUCLID
            else:
                metric = None
            from evaluate_retrieval import _resolve_kblocks
            _, top_i = sharded_topk(feats, feats[g0:g1], L, g0, metric=metric, group=group,
                                    local_topk=kernels.get('local_topk'), merge=kernels.get('merge'),
                                    kblocks=_resolve_kblocks(kblocks, int(feats.shape[1])))
            if q1 > q0:
                outs.append(kernels['hierarchical_precision'](top_i[q0:q1].contiguous(), cls_d, cls_d[q0:q1].contiguous(),
                                                             qidx_d[q0:q1].contiguous(), *args_d, ks_d, ahp_len=ahp_len, want_ap=False,
                                                             **curves(args_d)))
        else:
            args_d = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (wup_t, lcs_t)] + [best_w, best_l]
            extra = curves(args_d)      # once per gallery, shared by every tile
            # (16-bit ranks between the two kernels -- ranking_tiles(idx16=True), se_hierarchical_precision_r16 -- give the same results
            # from half the bytes, but measured at 50k x 50k the ranking gains 0.24 ms and the metric kernel, which is not bound by its
            # rank stream, loses 0.41 ms to the unpacking: int32 stays the default; SE_EVAL_IDX16=1 switches)
            tiles_kw = {'idx16': True} if (native_metrics and native_ranking and n <= 53248 and os.environ.get('SE_EVAL_IDX16')) else {}
            for r0, tile in kernels['ranking_tiles'](feats, normalize, tile_rows=tile_rows, queries=(q0, q1), kblocks=kblocks, **tiles_kw):
                rows = tile.shape[0]
                outs.append(kernels['hierarchical_precision'](tile, cls_d, cls_d[r0:r0 + rows].contiguous(), qidx_d[r0:r0 + rows].contiguous(),
                                                             *args_d, ks_d, ahp_len=ahp_len, want_ap=compute_ap, **extra))
        res_d = torch.cat(outs) if outs else torch.zeros((0, ncol), dtype=torch.float64, device=dev)
        sums = None
        if world > 1:
            if gather_per_query and per_query:    # ragged all-gather: pad every shard to the largest one
                rows_max = max(e - s for s, e in shard_bounds(n, world))
                padded = torch.zeros((rows_max, ncol), dtype=torch.float64, device=dev)
                padded[:res_d.shape[0]] = res_d
                gathered = torch.empty((world * rows_max, ncol), dtype=torch.float64, device=dev)
                dist.all_gather_into_tensor(gathered, padded, group=group)
                res_d = torch.cat([gathered[r * rows_max:r * rows_max + (e - s)] for r, (s, e) in enumerate(shard_bounds(n, world))])
                q0, q1 = 0, n
            else:                   # the means only: one all-reduce of ncol sums
                sums = res_d.sum(dim=0)
                dist.all_reduce(sums, group=group)
        nk = len(ks)
        my_ids = ids[q0:q1]
        prec = {}
        col = {}
        for t, k in enumerate(ks):
            col['P@{} (WUP)'.format(k)] = t
            col['P@{} (LCS_HEIGHT)'.format(k)] = nk + t
        if compute_ahp:
            sfx = '' if ahp_clip is None else '@{}'.format(ahp_clip)
            col['AHP{} (WUP)'.format(sfx)] = 2 * nk
            col['AHP{} (LCS_HEIGHT)'.format(sfx)] = 2 * nk + 1
        if compute_ap:
            col['AP'] = 2 * nk + 2
        if not per_query:
            if sums is None:        # every row is here (one process, or gathered): the column means
                sums = res_d.sum(dim=0)
            sums = sums.cpu().numpy()
            return {name: float(sums[c]) / n for name, c in col.items()}, None
        res = res_d.cpu().numpy()
        for name, c in col.items():
            prec[name] = dict(zip(my_ids, res[:, c].tolist()))
        if sums is not None:
            sums = sums.cpu().numpy()
            return {name: float(sums[c]) / n for name, c in col.items()}, prec
        return {metric: sum(values.values()) / len(values) for metric, values in prec.items()}, prec


def _average_precision(relevant):
    """AP of a ranking with distinct scores: mean over the relevant items of precision at their
    rank (== sklearn.metrics.average_precision_score for tie-free scores; 0 if nothing is relevant)."""
    hits = np.flatnonzero(relevant)
    if hits.size == 0:
        return 0.0
    return float(np.mean(np.arange(1, hits.size + 1) / (hits + 1.0)))
