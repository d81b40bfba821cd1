"""the retrieval meters (HitAtK / PrecisionAtK / RecallAtK / MeanAveragePrecisionAtK / NDCGAtK).

* the oracle (oracle/retrieval_ref.py) against the reference's OWN known-answer tables
  (tests/golden/retrieval_known_answers.npz = tests/base_tests/metrics/representation/data.py as data) and against
  tests/golden/retrieval_meters.npz (the reference's files driven on random data by tests/golden/gen_golden.py);
* the meters (torchok_amd/retrieval.py) against both, on the host stand-in and, marked gpu, through libtok_gfx950.so;
* the search kernels on their own against torch fp32, and a 4096-vector case against the oracle."""
import os

import numpy as np
import pytest
import torch

import oracle.retrieval_ref as R
import torchok_amd as T

GD = os.path.join(os.path.dirname(__file__), 'golden')
KNOWN = np.load(os.path.join(GD, 'retrieval_known_answers.npz'))
CASES = np.load(os.path.join(GD, 'retrieval_meters.npz'))
METERS = dict(hit_rate='HitAtKMeter', precision='PrecisionAtKMeter', recall='RecallAtKMeter',
              average_precision='MeanAveragePrecisionAtKMeter', ndcg='NDCGAtKMeter')


def _known_tables():
    for key in KNOWN.files:
        if key.startswith('answer__'):
            _, dataset, metric = key.split('__')
            yield dataset, metric, KNOWN[key]


def _known_kwargs(dataset):
    if dataset == 'classification':
        return dict(dataset_type='classification', normalize_vectors=True), dict(group_labels=KNOWN['targets'])
    scores = KNOWN['scores_query_as_relevant' if dataset == 'query_as_relevant' else 'scores']
    return (dict(dataset_type='representation', normalize_vectors=dataset == 'query_as_relevant'),
            dict(group_labels=KNOWN['group_labels'], query_idxs=KNOWN['queries_idx'], scores=scores))


@pytest.fixture(params=['host', pytest.param('hip', marks=pytest.mark.gpu)])
def dev(request):
    if request.param == 'host':
        request.getfixturevalue('fake_backend')
        return 'cpu'
    assert torch.cuda.is_available()
    return 'cuda'


# ---- oracle pins (CPU) ----------------------------------------------------------------------------------------------------
def test_oracle_reproduces_the_reference_known_answers():
    n = 0
    for dataset, metric, answers in _known_tables():
        ctor, data = _known_kwargs(dataset)
        for k in range(1, 7):
            v = R.meter_compute(metric, KNOWN['vectors'], ctor['dataset_type'], k=k,
                                normalize_vectors=ctor['normalize_vectors'], **data)
            np.testing.assert_almost_equal(v, answers[k - 1])      # the reference's own assertion (7 decimals)
            n += 1
    assert n == 54


def _case(c):
    metric, ds, k, dist, ga, katl = c.split('|')
    return metric, ds, int(k), dist, bool(int(ga)), bool(int(katl))


def test_oracle_reproduces_the_reference_files_on_random_data():
    for c, want in zip(CASES['cases'], CASES['expected']):
        metric, ds, k, dist, ga, katl = _case(str(c))
        gl = CASES['labels'] if ds == 'classification' else CASES['groups']
        v = R.meter_compute(metric, CASES['vectors'], ds, k=k, group_labels=gl, query_idxs=CASES['query_idxs'],
                            scores=CASES['scores'], metric_distance=dist, group_averaging=ga, k_as_target_len=katl)
        assert abs(v - want) < 1e-12, c


# ---- the meters ---------------------------------------------------------------------------------------------------------------
def _feed(meter, dev, vectors, batch=4, **data):
    n = len(vectors)
    for lo in range(0, n, batch):
        sl = slice(lo, lo + batch)
        meter.update(vectors=torch.from_numpy(vectors[sl]).to(dev),
                     **{k: torch.from_numpy(np.asarray(v)[sl]).to(dev) for k, v in data.items()})
    return meter.compute()


def test_meters_reproduce_the_reference_known_answers(dev):
    for dataset, metric, answers in _known_tables():
        ctor, data = _known_kwargs(dataset)
        for k in range(1, 7):
            m = T.METRICS.get(METERS[metric])(k=k, **ctor)
            np.testing.assert_almost_equal(_feed(m, dev, KNOWN['vectors'], batch=1, **data), answers[k - 1], decimal=6)


def test_meters_reproduce_the_reference_files_on_random_data(dev):
    for c, want in zip(CASES['cases'], CASES['expected']):
        metric, ds, k, dist, ga, katl = _case(str(c))
        m = T.METRICS.get(METERS[metric])(dataset_type=ds, k=k, metric_distance=dist, group_averaging=ga,
                                         k_as_target_len=katl, search_batch_size=7)
        if ds == 'classification':
            v = _feed(m, dev, CASES['vectors'], batch=10, group_labels=CASES['labels'])
        else:
            v = _feed(m, dev, CASES['vectors'], batch=10, group_labels=CASES['groups'], query_idxs=CASES['query_idxs'],
                      scores=CASES['scores'])
        assert abs(v - want) < 1e-6, (c, v, want)


def test_argument_errors(dev):
    M = T.METRICS.get('HitAtKMeter')
    with pytest.raises(KeyError):
        M(dataset_type='detection')
    with pytest.raises(KeyError):
        M(dataset_type='classification', metric_distance='cosine')
    with pytest.raises(NotImplementedError):
        M(dataset_type='classification', exact_index=False)
    m = M(dataset_type='classification')
    with pytest.raises(ValueError, match='group_labels must be not None'):
        m.update(vectors=torch.zeros(2, 4, device=dev))
    m.update(vectors=torch.randn(3, 4, device=dev), group_labels=torch.tensor([0, 0, 1], device=dev))
    with pytest.raises(ValueError, match='has only one element'):
        m.compute()
    m = M(dataset_type='classification', raise_empty_query=False)          # the lonely query scores 0
    m.update(vectors=torch.eye(3, 4, device=dev), group_labels=torch.tensor([0, 0, 1], device=dev))
    assert m.compute() == pytest.approx(2 / 3)
    r = T.METRICS.get('RecallAtKMeter')(dataset_type='representation')
    with pytest.raises(ValueError, match='scores must be not None'):
        r.update(vectors=torch.zeros(2, 4, device=dev), query_idxs=torch.tensor([0, -1], device=dev))
    m.reset()
    with pytest.raises(RuntimeError):
        m.compute()


# ---- kernels on their own (GPU) ----------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize('nq,ng,d,metric', [(70, 130, 24, 0), (1, 65, 7, 1), (129, 64, 200, 1), (300, 1000, 128, 0)])
def test_sim_matrix_and_topk_kernels(nq, ng, d, metric):
    from torchok_amd import _C
    from torchok_amd.engine.core import ptr, stream_ptr
    g = torch.Generator().manual_seed(nq + ng)
    q, ga = torch.randn(nq, d, generator=g).cuda(), torch.randn(ng, d, generator=g).cuda()
    out = torch.empty(nq, ng, device='cuda')
    _C.check(_C.lib().tok_sim_matrix(ptr(q), ptr(ga), nq, ng, d, d, d, metric, ptr(out), ng, stream_ptr()), 'sim')
    want = q.double() @ ga.double().t() if metric == 0 else -torch.cdist(q.double(), ga.double()) ** 2
    assert (out.double() - want).abs().max() < 1e-4 * max(1.0, float(want.abs().max()))
    for k in (1, 5, ng + 3):
        vals = torch.empty(nq, k, device='cuda')
        idx = torch.empty(nq, k, dtype=torch.int64, device='cuda')
        _C.check(_C.lib().tok_topk_rows(ptr(out), nq, ng, ng, k, ptr(vals), ptr(idx), stream_ptr()), 'topk')
        kk = min(k, ng)
        order = torch.argsort(-out, dim=1, stable=True)[:, :kk]
        assert torch.equal(idx[:, :kk], order) and torch.equal(vals[:, :kk], torch.gather(out, 1, order))
        assert (idx[:, kk:] == -1).all() and torch.isinf(vals[:, kk:]).all()


@pytest.mark.gpu
def test_topk_ties_take_the_lower_index_and_nan_never_ranks():
    from torchok_amd import _C
    from torchok_amd.engine.core import ptr, stream_ptr
    s = torch.tensor([[1., 3., 3., float('nan'), 2., 3.], [0., 0., 0., 0., 0., 0.]], device='cuda')
    vals = torch.empty(2, 5, device='cuda')
    idx = torch.empty(2, 5, dtype=torch.int64, device='cuda')
    _C.check(_C.lib().tok_topk_rows(ptr(s), 2, 6, 6, 5, ptr(vals), ptr(idx), stream_ptr()), 'topk')
    assert idx.tolist() == [[1, 2, 5, 4, 0], [0, 1, 2, 3, 4]]


@pytest.mark.gpu
@pytest.mark.parametrize('metric,kw', [('hit_rate', dict(k=1)), ('recall', dict(k=4, metric_distance='L2')),
                                       ('average_precision', dict(k=10, group_averaging=True)),
                                       ('ndcg', dict(k=5, normalize_vectors=True))])
def test_four_thousand_vectors_against_the_oracle(metric, kw):
    rng = np.random.default_rng(11)
    n, d, classes = 4096, 128, 256
    centers = rng.standard_normal((classes, d)).astype(np.float32)
    labels = rng.permutation(np.repeat(np.arange(classes), n // classes))
    vectors = (centers[labels] + 1.5 * rng.standard_normal((n, d))).astype(np.float32)
    m = T.METRICS.get(METERS[metric])(dataset_type='classification', **kw)
    got = _feed(m, 'cuda', vectors, batch=512, group_labels=labels)
    want = R.meter_compute(metric, vectors, 'classification', group_labels=labels, **kw)
    assert 0.05 < want < 0.99 and abs(got - want) < 2e-4, (got, want)     # a near-tie may swap two neighbours
