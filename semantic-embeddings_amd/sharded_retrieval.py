"""Sharded-gallery retrieval across the GPUs of one node (BASELINE.json configs[4], SURVEY.md
section 8e row 3; not present in the reference, which ranks on one host).

Every rank holds ``1/G`` of the gallery rows (plus their global row offset) and ALL queries.  Per
step:  local fused distance + top-k over the shard  ->  one RCCL all-gather of the per-shard
``(distance f32, global index i32)`` lists  ->  k-way merge under the canonical (distance, index)
order.  Because ties are broken by the GLOBAL index the result does not depend on G.

The all-gather moves ``Q * k * 8`` bytes per rank (100 MB at Q = 50k, k = 250); xGMI is a full
mesh, so the direct all-gather keeps all 7 links of a GPU busy at once.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """Row ranges of an n-row gallery split into ``world`` near-equal contiguous shards."""
    base, extra = divmod(n, world)
    bounds, start = [], 0
    for r in range(world):
        size = base + (1 if r < extra else 0)
        bounds.append((start, start + size))
        start += size
    return bounds


def sharded_topk(queries, gallery_shard, k, shard_offset, metric=None, group=None, local_topk=None, merge=None, kblocks=None):
    """Global top-k of ``queries`` against the gallery whose local shard is ``gallery_shard``.

    Returns ``(dist [Q, k] f32, idx [Q, k] i32)`` identical on every rank.  ``local_topk`` /
    ``merge`` default to the HIP kernels (``sehip.retrieve_topk`` / ``sehip.topk_merge``); tests
    inject CPU stand-ins to exercise the collective logic under gloo.  ``kblocks``: the BLAS K-block
    list of the distance arithmetic (D > 448, see ``evaluate_retrieval.host_blas_kblocks``); it reaches
    every rank's local kernel (and an injected ``local_topk`` as its fifth argument) so that the sharded
    result equals the single-process one bit for bit.

    Exchange: ONE all-gather of the packed ``[2, Q, k]`` (distance bits | indices) block of every rank; the HIP
    kernels write their lists straight into the send block and merge straight out of the receive buffer."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    k_local = min(k, gallery_shard.shape[0])
    kb = list(kblocks) if kblocks is not None and len(kblocks) > 1 else None
    packed = None
    if local_topk is None:
        import sehip
        metric = sehip.METRIC_COSINE if metric is None else metric
        out = None
        if world > 1 and k_local == k:         # lists land in the all-gather send block: no packing copy
            packed = torch.empty((2, queries.shape[0], k), dtype=torch.int32, device=queries.device)
            out = (packed[0].view(torch.float32), packed[1])
        d, i = sehip.retrieve_topk(queries, gallery_shard, k_local, metric=metric, col_offset=shard_offset, kblocks=kb, out=out)
    elif kb is not None:
        d, i = local_topk(queries, gallery_shard, k_local, shard_offset, kb)
    else:
        d, i = local_topk(queries, gallery_shard, k_local, shard_offset)
    if k_local < k:   # tiny shard: pad with +inf so every rank contributes [Q, k]
        pad = k - k_local
        d = torch.cat([d, torch.full((d.shape[0], pad), float('inf'), dtype=d.dtype, device=d.device)], dim=1)
        i = torch.cat([i, torch.full((i.shape[0], pad), 2 ** 31 - 1, dtype=i.dtype, device=i.device)], dim=1)
    if world == 1:
        return d, i
    gathered = all_gather_packed(pack_lists(d, i) if packed is None else packed, world, group)
    if merge is None:
        import sehip
        return sehip.topk_merge(gathered)                                  # se_topk_merge_packed: reads the receive buffer in place
    return merge(gathered[:, 0].view(torch.float32), gathered[:, 1])


def pack_lists(d, i):
    """(dist f32 [Q, k], idx i32 [Q, k]) -> ONE int32 buffer [2, Q, k]: the distance bits, then the indices."""
    packed = torch.empty((2,) + tuple(d.shape), dtype=torch.int32, device=d.device)
    packed[0].copy_(d.contiguous().view(torch.int32))
    packed[1].copy_(i)
    return packed


def all_gather_packed(packed, world, group=None):
    """The exchange step of the sharded-gallery split: every rank's packed ``[2, Q, k]`` block -> ``[world, 2, Q, k]``
    (rank-major) by ONE all-gather of ``8 Q k`` bytes per rank instead of two of half the size (a direct all-gather over the
    xGMI mesh pays its launch and latency once)."""
    if world == 1:
        return packed.unsqueeze(0)
    return _all_gather_rows(packed.view(1, -1), world, group).view((world,) + tuple(packed.shape))


def _all_gather_rows(t, world, group=None):
    """[rows, k] per rank -> [world * rows, k].  RCCL ("nccl") gathers device tensors directly over xGMI; the gloo backend
    (CPU tests, or two test processes sharing one GPU -- RCCL refuses two ranks on one device) has no device all-gather, so
    the lists take the host route there."""
    if t.is_cuda and dist.get_backend(group) == 'gloo':
        host = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype)
        dist.all_gather_into_tensor(host, t.cpu(), group=group)
        return host.to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out
