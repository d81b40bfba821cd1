"""CPU restatement of the retrieval meters of the validation path: numpy, fp32 vectors,
each function citing the reference lines it follows.  TEST INFRASTRUCTURE ONLY.

Third-party pieces the reference calls and that are absent here, restated from their published behaviour:
  * faiss-cpu 1.7.2 (pyproject.toml:26) `IndexFlatIP` / `IndexFlatL2` with `exact_index=True`: exhaustive search,
    scores = inner products (descending) resp. squared Euclidean distances (ascending)  -> `flat_search`.
  * ranx 0.3.8 (poetry.lock:2597-2598) `hit_rate`, `precision`, `recall`, `average_precision`, `ndcg` (Jarvelin,
    linear gains; relevance level 1; `k` clipped to the run length)  -> `RANX`.

Pinned by (a) the reference's OWN known-answer tables for these meters
(`tests/base_tests/metrics/representation/data.py`: CLASSIFICATION_ANSWERS, REPRESENTATION_ANSWERS,
REPRESENTATION_QUERY_AS_RELEVANT_ANSWERS), copied as data into tests/golden/retrieval_known_answers.npz, and
(b) tests/golden/retrieval_meters.npz, which tests/golden/gen_golden.py writes by driving the reference's own
`prepare_classification_data` / `prepare_representation_data` / `query_generator` / `clear_faiss_output` /
`process_data_for_metric_func` (index_base_metric.py, representation_ranx.py) on random data.

Note on index_base_metric.py:190: as written it divides by `np.linalg.norm(vectors, axis=0)` (per-COLUMN norms); the
reference's known-answer tables (and the "IP - cosine distance" contract of the docstring, :66) hold only for per-ROW
(unit-vector) normalisation — with the literal axis the reference's own test_precision_when_dataset_is_classification
gives 2/9 instead of 4/9.  The restatement follows the tests: unit-length rows.
Note on index_base_metric.py:262: `min(args[0].shape)` is evaluated on the python list the Ranx meters return, so
`compute()` of the reference as shipped (v0.4.12) stops there; the restatement follows the evident intent of that
line (skip a search batch that holds no relevant entries) and the known-answer tables above."""
import math

import numpy as np


# ---- faiss IndexFlat{IP,L2}.search ------------------------------------------------------------------------------------
def flat_search(gallery: np.ndarray, queries: np.ndarray, k: int, metric: str = 'IP'):
    """(scores [nq][k], indices [nq][k]) of the k best gallery rows per query; ties -> lower index first.  A label of
    -1 then indexes `faiss_vector_idxs[-1]` in query_generator (:503), i.e. the LAST gallery row — kept as is."""
    g = gallery.astype(np.float32)
    q = queries.astype(np.float32)
    if metric == 'IP':
        s = q @ g.T
        order = np.argsort(-s, axis=1, kind='stable')[:, :k]
    else:
        s = ((q[:, None, :] - g[None, :, :]) ** 2).sum(-1)
        order = np.argsort(s, axis=1, kind='stable')[:, :k]
    val = np.take_along_axis(s, order, axis=1)
    if k > g.shape[0]:      # faiss pads missing results with label -1 (score -inf for IP, +inf for L2)
        pad = k - g.shape[0]
        order = np.concatenate([order, np.full((len(q), pad), -1, dtype=order.dtype)], axis=1)
        val = np.concatenate([val, np.full((len(q), pad), -np.inf if metric == 'IP' else np.inf, np.float32)], axis=1)
    return val, order


# ---- ranx 0.3.8 metrics on one query ---------------------------------------------------------------------------------------
def _clean(qrels):                       # relevance level 1
    return [(d, g) for d, g in qrels if g >= 1]


def _fix_k(k, run):
    return len(run) if (k == 0 or k > len(run)) else k


def _hit_list(qrels, run, k):
    rel = {d for d, _ in qrels}
    return [1.0 if d in rel else 0.0 for d in run[:k]]


def hit_rate(qrels, run, k):
    qrels = _clean(qrels)
    if not qrels:
        return 0.0
    return 1.0 if sum(_hit_list(qrels, run, _fix_k(k, run))) > 0 else 0.0


def precision(qrels, run, k):
    qrels = _clean(qrels)
    if not qrels:
        return 0.0
    k = _fix_k(k, run)
    return sum(_hit_list(qrels, run, k)) / k


def recall(qrels, run, k):
    qrels = _clean(qrels)
    if not qrels:
        return 0.0
    return sum(_hit_list(qrels, run, _fix_k(k, run))) / len(qrels)


def average_precision(qrels, run, k):
    qrels = _clean(qrels)
    if not qrels:
        return 0.0
    hits = _hit_list(qrels, run, _fix_k(k, run))
    acc, seen = 0.0, 0.0
    for i, h in enumerate(hits):
        seen += h
        acc += h * seen / (i + 1)
    return acc / len(qrels)


def ndcg(qrels, run, k):
    qrels = _clean(qrels)
    if not qrels:
        return 0.0
    gain = dict(qrels)
    kk = _fix_k(k, run)
    dcg = sum(gain.get(d, 0.0) / math.log2(i + 2) for i, d in enumerate(run[:kk]))
    ideal = sorted((g for _, g in qrels), reverse=True)
    ki = len(ideal) if (k == 0 or k > len(ideal)) else k
    idcg = sum(g / math.log2(i + 2) for i, g in enumerate(ideal[:ki]))
    return dcg / idcg


RANX = dict(hit_rate=hit_rate, precision=precision, recall=recall, average_precision=average_precision, ndcg=ndcg)


# ---- IndexBasedMeter ------------------------------------------------------------------------------------------------------
def prepare_classification(targets: np.ndarray, raise_empty_query=True):
    """index_base_metric.py:379-418: every vector is a query; relevant = same label, the query itself dropped.
    Queries come grouped by label in ascending label order (pandas groupby), rows ascending inside a group."""
    relevant, rows = [], []
    for lab in np.unique(targets):
        group = np.where(targets == lab)[0]
        for qi in group:
            rel = [int(j) for j in group if j != qi]
            if not rel and raise_empty_query:
                raise ValueError(f'Representation metric. The class {lab} has only one element.')
            rows.append(int(qi))
            relevant.append(rel)
    n = len(targets)
    return relevant, np.arange(n), np.asarray(rows), np.ones(n, dtype=bool)


def prepare_representation(query_idxs: np.ndarray, scores: np.ndarray, raise_empty_query=True):
    """index_base_metric.py:342-377."""
    is_query = query_idxs >= 0
    q_cols = query_idxs[is_query]
    q_rows = np.where(is_query)[0]
    q_as_rel = np.any(scores[q_rows, :] > 0, axis=-1)
    gallery = np.delete(np.arange(len(scores)), q_rows[~q_as_rel])
    relevant = []
    for c in q_cols:
        idx = np.where(scores[:, c] > 0.0)[0]
        if len(idx) == 0:
            if raise_empty_query:
                raise ValueError('Representation metric. The dataset contains a query vector that does not '
                                 'has relevants. Set parameter raise_empty_query to False for compute.')
            relevant.append([])
        else:
            order = np.argsort(scores[idx, c])
            relevant.append([int(j) for j in idx[order[::-1]]])
    return relevant, gallery, q_cols, q_rows, q_as_rel


def meter_compute(metric: str, vectors, dataset_type: str, k=None, group_labels=None, query_idxs=None, scores=None,
                  metric_distance='IP', normalize_vectors=False, group_averaging=False, k_as_target_len=False,
                  raise_empty_query=True, per_query=False):
    """IndexBasedMeter.compute (index_base_metric.py:170-270) with RanxBasedMeter.process_data_for_metric_func
    (representation_ranx.py:29-53).  Search batching (`search_batch_size`) only splits the work; the mean over queries
    is independent of it, so it is not restated."""
    fn = RANX[metric]
    k = 1 if k is None else k                                             # :104
    search_k = k + 1                                                      # :107
    vectors = np.array(vectors, dtype=np.float32)
    if normalize_vectors:
        vectors = vectors / np.linalg.norm(vectors, axis=1, keepdims=True)   # :189-190, see the header note
    group_labels = np.asarray(group_labels)
    if dataset_type == 'classification':
        relevant, gallery, q_rows, q_as_rel = prepare_classification(group_labels, raise_empty_query)
        q_cols = None
    else:
        scores = np.asarray(scores)
        relevant, gallery, q_cols, q_rows, q_as_rel = prepare_representation(np.asarray(query_idxs), scores,
                                                                             raise_empty_query)
    if group_averaging:                                                   # :222-228
        groups = [np.where(group_labels == lab)[0] for lab in np.unique(group_labels)]
    else:
        groups = [np.arange(len(group_labels))]
    values, per_q = [], {}
    for g in groups:
        sel = np.where(np.isin(q_rows, g))[0]                             # :234
        kk = (len(g) + 1 - int((~q_as_rel[sel]).sum())) if k_as_target_len else search_k   # :240-245
        if len(sel) == 0:
            values.append(float('nan'))
            continue
        _, local = flat_search(vectors[gallery], vectors[q_rows[sel]], kk, metric_distance)
        closest = gallery[local]
        total = 0.0
        for n_, qi in enumerate(sel):
            run = closest[n_][1:] if q_as_rel[qi] else closest[n_][:-1]   # clear_faiss_output :420-444
            if q_cols is None:
                qrels = [(j, 1.0) for j in relevant[qi]]
            else:
                qrels = [(j, float(scores[j, q_cols[qi]])) for j in relevant[qi]]
            v = fn(qrels, [int(d) for d in run], kk - 1)
            per_q[int(q_rows[qi])] = v
            total += v
        values.append(total / len(sel))                                   # :266
    out = float(np.mean(values))                                          # :269
    return (out, per_q) if per_query else out
