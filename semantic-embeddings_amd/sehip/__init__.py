"""sehip -- Python binding of libsehip.so, the MI355X (gfx950) kernels behind the cosine-embedding
training + retrieval hot path.  See include/sehip.h for the C ABI and DESIGN.md for the design."""
from ._lib import (DTYPE_BF16, DTYPE_F32, EXPORTS, LIB_PATH, METRIC_COSINE, METRIC_DOT, METRIC_EUCLID, TOPK_MAX,
                   SehipError, build, lib)
from .ops import (empty_rows, hierarchical_precision, hprec_reciprocal_curves, cosine_embedding_loss, devise_ranking_loss, cosine_loss_backward, cosine_loss_forward, l2norm, labelembed_loss, nn_accuracy, normalize_rows_,
                  pairwise_dist, rank_rows, rank_rows_check, rank_rows_init, rank_rows_workspace_bytes, release_workspace, retrieve_topk, row_sqnorm,
                  topk_merge, topk_rows, workspace_bytes, squared_distance_loss, sqdist_loss_forward, sqdist_loss_backward,
                  phase_timing, phase_timing_read)
from . import ops
