"""Retrieval meters of the validation path: ``HitAtKMeter``, ``PrecisionAtKMeter``, ``RecallAtKMeter``,
``MeanAveragePrecisionAtKMeter``, ``NDCGAtKMeter`` with the constructor surface, ``update`` arguments and ``compute``
semantics of the reference (``torchok/metrics/index_base_metric.py:32-270``, ``metrics/representation_ranx.py:19-123``),
the way ``pairwise_sop.yaml:145-155`` / ``triplet_sop.yaml`` / ``representation_arcface_sop.yaml`` configure them.

The reference stores the vectors, copies them to the host, builds a faiss flat index and calls ranx per search batch.
Here the vectors never leave HBM: exhaustive search = ``tok_sim_matrix`` (exact fp32) + ``tok_topk_rows`` per query chunk,
relevance and the ranking metric are evaluated per query by ``tok_retrieval_nrel`` / ``tok_retrieval_eval``, the mean is a
fixed-order fp64 fold (``tok_colsum_f32``); only the index bookkeeping of ``prepare_*_data`` (labels, query ids — integers)
runs on the host, and one float per group comes back.

Differences, all deliberate:
  * ``normalize_vectors`` makes unit-length ROWS (what the reference's known-answer tests and its "IP = cosine" contract
    require); the shipped line (:190) divides by per-column norms.
  * ``exact_index=False`` (faiss IVF, approximate) is not provided.
  * ``search_batch_size`` / ``use_batching_search`` only chunk the faiss requests in the reference (the result does not
    depend on them); accepted and ignored — the chunk is chosen from the size of the similarity matrix.
  * the torchmetrics-backed ``Retrieval*Meter`` family (representation_torchmetrics.py) is not provided."""
from typing import List, Optional

import numpy as np
import torch
from torch import Tensor, nn

from . import _C
from .constructor import METRICS
from .engine.core import ptr, require_device, stream_ptr

F32 = torch.float32
_KIND = dict(hit_rate=0, precision=1, recall=2, average_precision=3, ndcg=4)
_SIM_BYTES = 1 << 30          # similarity-matrix chunk kept under 1 GiB of HBM


class IndexBasedMeter(nn.Module):
    """index_base_metric.py:32-120 (states, argument checks) and :170-270 (compute)."""

    def __init__(self, exact_index: bool, dataset_type: str, metric_distance: str, metric_func: str,
                 k_as_target_len: bool = False, k: Optional[int] = None, use_batching_search: bool = True,
                 search_batch_size: Optional[int] = None, normalize_vectors: bool = False, group_averaging: bool = False,
                 raise_empty_query: bool = True, **kwargs):
        super().__init__()
        if dataset_type not in ('classification', 'representation'):
            raise KeyError(dataset_type)                       # dataset_enum_mapping[...] (:93)
        if metric_distance not in ('IP', 'L2'):
            raise KeyError(metric_distance)                    # distance_enum_mapping[...] (:94)
        if not exact_index:
            raise NotImplementedError('torchok_amd retrieval meters: exact_index=True only (exhaustive search on the GPU)')
        self.dataset_type, self.metric_distance, self.metric_func = dataset_type, metric_distance, metric_func
        self.normalize_vectors, self.group_averaging = normalize_vectors, group_averaging
        self.k_as_target_len, self.raise_empty_query = k_as_target_len, raise_empty_query
        self.use_batching_search, self.search_batch_size = use_batching_search, search_batch_size
        k = 1 if k is None else k                               # :104
        self.search_k, self.metric_compute_k = k + 1, k        # :107-109
        self.reset()

    def reset(self) -> None:
        self.vectors: List[Tensor] = []
        self.group_labels: List[Tensor] = []
        self.query_idxs: List[Tensor] = []
        self.scores: List[Tensor] = []

    # ---- update (:122-168) ----------------------------------------------------------------------------------------------------
    def update(self, vectors: Tensor, group_labels: Optional[Tensor] = None, query_idxs: Optional[Tensor] = None,
               scores: Optional[Tensor] = None) -> None:
        require_device(vectors)
        self.vectors.append(vectors.detach())
        if self.dataset_type == 'classification':
            if group_labels is None:
                raise ValueError('In classification dataset group_labels must be not None.')
            self.group_labels.append(group_labels.detach())
        else:
            if query_idxs is None:
                raise ValueError('In representation dataset query_numbers must be not None.')
            if scores is None:
                raise ValueError('In representation dataset scores must be not None')
            self.query_idxs.append(query_idxs.detach())
            self.scores.append(scores.detach())
            self.group_labels.append(group_labels.detach())

    # ---- state sync: dist_reduce_fx="cat" (:112-120) -----------------------------------------------------------------------------
    @staticmethod
    def _cat_over_ranks(t: Tensor) -> Tensor:
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return t
        world = dist.get_world_size()
        n = torch.tensor([t.shape[0]], device=t.device, dtype=torch.int64)
        sizes = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(sizes, n)
        sizes = [int(s) for s in sizes]
        pad = torch.zeros((max(sizes),) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[:t.shape[0]] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        return torch.cat([p[:s] for p, s in zip(parts, sizes)])

    # ---- host bookkeeping (integers only) ------------------------------------------------------------------------------------------
    def _prepare_classification(self, targets: np.ndarray):
        """:379-418 — queries grouped by label (ascending), rows ascending inside a group; everything is in the index."""
        order = np.argsort(targets, kind='stable')
        if self.raise_empty_query:
            labs, counts = np.unique(targets, return_counts=True)
            if (counts == 1).any():
                raise ValueError(f'Representation metric. The class {labs[counts == 1][0]} has only one element.')
        n = len(targets)
        return None, order.astype(np.int64), None, np.ones(n, dtype=bool)

    def _prepare_representation(self, query_idxs: np.ndarray, scores: np.ndarray):
        """:342-377."""
        is_query = query_idxs >= 0
        q_cols = query_idxs[is_query].astype(np.int64)
        q_rows = np.where(is_query)[0].astype(np.int64)
        q_as_rel = np.any(scores[q_rows, :] > 0, axis=-1)
        gallery = np.delete(np.arange(len(scores), dtype=np.int64), q_rows[~q_as_rel])
        if self.raise_empty_query and not np.all((scores[:, q_cols] > 0).any(axis=0)):
            raise ValueError('Representation metric. The dataset contains a query vector that does not '
                             'has relevants. Set parameter raise_empty_query to False for compute.')
        return gallery, q_rows, q_cols, q_as_rel

    # ---- compute (:170-270) ------------------------------------------------------------------------------------------------------------
    def compute(self) -> float:
        if not self.vectors:
            raise RuntimeError(f'{type(self).__name__}.compute() before any update()')
        lib, st = _C.lib(), stream_ptr()
        vec = self._cat_over_ranks(torch.cat(self.vectors).to(F32).contiguous())
        dev = vec.device
        n, d = vec.shape
        if self.normalize_vectors:
            unit, inv = torch.empty_like(vec), torch.empty(n, dtype=F32, device=dev)
            _C.check(lib.tok_l2norm_fwd(ptr(vec), ptr(unit), ptr(inv), n, d, d, 1, 0.0, st), 'tok_l2norm_fwd')
            vec = unit
        labels_t = self._cat_over_ranks(torch.cat(self.group_labels).to(torch.int64).contiguous())
        group_labels = labels_t.cpu().numpy()
        scores_t = ideal_src = None
        if self.dataset_type == 'classification':
            gallery, q_rows, q_cols, q_as_rel = self._prepare_classification(group_labels)
            gal_vec = vec
        else:
            scores_t = self._cat_over_ranks(torch.cat(self.scores).to(F32).contiguous())
            query_idxs = self._cat_over_ranks(torch.cat(self.query_idxs).to(torch.int64)).cpu().numpy()
            gallery, q_rows, q_cols, q_as_rel = self._prepare_representation(query_idxs, scores_t.cpu().numpy())
            gal_vec = vec[torch.from_numpy(gallery).to(dev)].contiguous()
            ideal_src = scores_t.t().contiguous()              # [query column][row]: rows of gains to rank for IDCG
        ng = gal_vec.shape[0]
        if self.group_averaging:                               # :222-228
            groups = [np.where(group_labels == lab)[0] for lab in np.unique(group_labels)]
        else:
            groups = [None]
        kind = _KIND[self.metric_func]
        values = []
        for g in groups:
            sel = np.arange(len(q_rows)) if g is None else np.where(np.isin(q_rows, g))[0]     # :234
            if len(sel) == 0:
                values.append(float('nan'))                    # 0 / 0 at :266
                continue
            if self.k_as_target_len:                           # :240-243
                kk = (n if g is None else len(g)) + 1 - int((~q_as_rel[sel]).sum())
            else:
                kk = self.search_k
            if kk < 2:
                values.append(0.0)
                continue
            nq = len(sel)
            rows_t = torch.from_numpy(q_rows[sel]).to(dev)
            cols_t = torch.from_numpy(q_cols[sel]).to(dev) if q_cols is not None else None
            drop_t = torch.from_numpy(q_as_rel[sel].astype(np.uint8)).to(dev)
            gal_t = torch.from_numpy(gallery).to(dev) if gallery is not None else None
            n_rel = torch.empty(nq, dtype=torch.int32, device=dev)
            cls = self.dataset_type == 'classification'
            _C.check(lib.tok_retrieval_nrel(ptr(labels_t) if cls else None, None if cls else ptr(scores_t), n,
                                            0 if cls else scores_t.shape[1], ptr(rows_t), ptr(cols_t), nq, ptr(n_rel), st),
                     'tok_retrieval_nrel')
            ideal = None
            if kind == 4 and not cls:
                ideal = torch.empty((nq, kk - 1), dtype=F32, device=dev)
                ii = torch.empty((nq, kk - 1), dtype=torch.int64, device=dev)
                src = ideal_src[cols_t].contiguous()
                _C.check(lib.tok_topk_rows(ptr(src), nq, n, n, kk - 1, ptr(ideal), ptr(ii), st), 'tok_topk_rows')
            idx = torch.empty((nq, kk), dtype=torch.int64, device=dev)
            val = torch.empty((nq, kk), dtype=F32, device=dev)
            chunk = max(1, min(nq, _SIM_BYTES // (4 * ng), 65535 * 64))
            sim = torch.empty((min(chunk, nq), ng), dtype=F32, device=dev)
            qv = vec[rows_t].contiguous()
            for lo in range(0, nq, chunk):
                m = min(chunk, nq - lo)
                _C.check(lib.tok_sim_matrix(ptr(qv[lo:]), ptr(gal_vec), m, ng, d, d, d, 0 if self.metric_distance == 'IP' else 1,
                                            ptr(sim), ng, st), 'tok_sim_matrix')
                _C.check(lib.tok_topk_rows(ptr(sim), m, ng, ng, kk, ptr(val[lo:]), ptr(idx[lo:]), st), 'tok_topk_rows')
            per_q = torch.empty(nq, dtype=F32, device=dev)
            _C.check(lib.tok_retrieval_eval(kind, ptr(idx), kk, ptr(drop_t), ptr(gal_t), ng, ptr(labels_t) if cls else None,
                                            None if cls else ptr(scores_t), 0 if cls else scores_t.shape[1], ptr(rows_t),
                                            ptr(cols_t), ptr(n_rel), ptr(ideal), nq, ptr(per_q), st), 'tok_retrieval_eval')
            total = torch.empty(1, dtype=F32, device=dev)
            _C.check(lib.tok_colsum_f32(ptr(per_q), nq, 1, ptr(total), 0, st), 'tok_colsum_f32')
            values.append(float(total) / nq)                   # :266
        return float(np.mean(values))                          # :269

    def forward(self, *args, **kwargs):
        self.update(*args, **kwargs)


def _meter(name: str, func: str):
    def __init__(self, dataset_type: str, exact_index: bool = True, metric_distance: str = 'IP', k: Optional[int] = None,
                 search_batch_size: Optional[int] = None, normalize_vectors: bool = False, group_averaging: bool = False,
                 k_as_target_len: bool = False, use_batching_search: bool = True, raise_empty_query: bool = True, **kwargs):
        IndexBasedMeter.__init__(self, exact_index=exact_index, dataset_type=dataset_type, metric_distance=metric_distance,
                                 metric_func=func, k=k, search_batch_size=search_batch_size,
                                 normalize_vectors=normalize_vectors, group_averaging=group_averaging,
                                 k_as_target_len=k_as_target_len, use_batching_search=use_batching_search,
                                 raise_empty_query=raise_empty_query, **kwargs)
    cls = type(name, (IndexBasedMeter,), {'__init__': __init__, '__module__': __name__,
                                          '__doc__': f'representation_ranx.py: ranx `{func}` over the exhaustive search.'})
    return METRICS.register_class(cls)


HitAtKMeter = _meter('HitAtKMeter', 'hit_rate')                                   # representation_ranx.py:56-67
PrecisionAtKMeter = _meter('PrecisionAtKMeter', 'precision')                      # :70-81
RecallAtKMeter = _meter('RecallAtKMeter', 'recall')                               # :84-95
MeanAveragePrecisionAtKMeter = _meter('MeanAveragePrecisionAtKMeter', 'average_precision')   # :98-109
NDCGAtKMeter = _meter('NDCGAtKMeter', 'ndcg')                                     # :112-123
