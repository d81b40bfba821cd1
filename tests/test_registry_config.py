"""Host-logic tests (CPU): registries and YAML config — the drop-in boundary of
reference torchok/constructor/registry.py:10-138, constructor/__init__.py:4-17, config_structure.py."""
import os
import textwrap

import pytest

import torchok_amd as T
from torchok_amd.constructor import registry as reg_mod
from torchok_amd.constructor.config import load_config
from torchok_amd.constructor.registry import Registry


def test_fourteen_registries_exist():
    import torchok_amd.constructor as C
    names = ['DATASETS', 'TRANSFORMS', 'OPTIMIZERS', 'SCHEDULERS', 'LOSSES', 'METRICS', 'CALLBACKS', 'TASKS',
             'BACKBONES', 'POOLINGS', 'HEADS', 'NECKS', 'DETECTION_NECKS', 'SAMPLERS']
    for n in names:
        assert isinstance(getattr(C, n), Registry)


def test_registry_semantics():
    r = Registry('things')

    @r.register_class
    class Foo:
        pass

    assert r.get('Foo') is Foo and r['Foo'] is Foo and 'Foo' in r
    with pytest.raises(KeyError):          # reference registry.py:57-58
        r.get('Bar')
    with pytest.raises(KeyError):          # duplicate, :80-81
        r.register_class(Foo)
    with pytest.raises(TypeError):         # non-callable, :76-77
        r.register_class(3)
    assert 'Foo' in __import__(Foo.__module__, fromlist=['x']).__all__


def test_list_models_natural_order_and_filters():
    names = T.BACKBONES.list_models('resnet*')
    assert names == ['resnet18', 'resnet18d', 'resnet26', 'resnet26d', 'resnet26t', 'resnet34', 'resnet34d', 'resnet50',
                     'resnet50d', 'resnet50t', 'resnet101', 'resnet101d', 'resnet152', 'resnet152d', 'resnet200',
                     'resnet200d']                                               # natural, not lexical
    assert T.BACKBONES.list_models('resnet*', exclude_filters=['resnet1*', 'resnet2*', '*d', '*t']) == ['resnet34', 'resnet50']
    assert reg_mod._natural_key('resnet101') == ['resnet', 101, '']


def test_identity_loss_passes_a_precomputed_loss_through():
    import torch
    loss = T.LOSSES.get('Identity')()
    x = torch.tensor(1.5)
    assert loss(x) is x


def test_hot_path_names_registered():
    for reg, names in ((T.BACKBONES, ['resnet18', 'resnet50']), (T.POOLINGS, ['Pooling', 'PoolingLinear']),
                       (T.HEADS, ['LinearHead', 'ClassificationHead']), (T.LOSSES, ['CrossEntropyLoss']),
                       (T.OPTIMIZERS, ['SGD', 'Adam', 'AdamW']), (T.TASKS, ['ClassificationTask']),
                       (T.SCHEDULERS, ['ExponentialLR', 'ReduceLROnPlateau'])):
        for n in names:
            assert n in reg, n


CIFAR_LIKE = textwrap.dedent('''
    task:
      name: ClassificationTask
      params:
        backbone_name: resnet18
        backbone_params:
          pretrained: false
          in_channels: 3
        pooling_name: Pooling
        head_name: ClassificationHead
        head_params:
          num_classes: &num_classes 10
        inputs:
          - shape: [3, &height 32, &width 32]
            dtype: &input_dtype float16
    joint_loss:
      losses:
        - name: CrossEntropyLoss
          mapping:
              input: prediction
              target: target
    optimization:
      - optimizer:
          name: Adam
          params:
            lr: 0.0001
        scheduler:
          name: ExponentialLR
          params:
            gamma: 0.97
    data:
      TRAIN:
        - dataloader:
            batch_size: 128
          dataset:
            name: CIFAR10
            params:
              data_folder: &data_folder ${oc.env:HOME}/.cache/torchok/cifar10/data
    trainer:
      accelerator: 'gpu'
      precision: 16
    logger:
      log_dir: '${oc.env:HOME}/.cache/torchok/cifar10/logs'
      experiment_name: resnet18
      timestamp: '${now:%Y-%m-%d}'
      name: TensorBoardLogger
    hydra:
      run:
        dir: &logs_dir '${logger.log_dir}/${logger.experiment_name}/${logger.timestamp}'
    callbacks:
      - name: ModelCheckpoint
        params:
          dirpath: *logs_dir
''')


def test_yaml_config_drives_the_task(tmp_path):
    """The keys/anchors/interpolations of examples/configs/classification_cifar10.yaml."""
    p = tmp_path / 'cfg.yaml'
    p.write_text(CIFAR_LIKE)
    cfg = load_config(str(p))
    home = os.environ['HOME']
    assert cfg.data['TRAIN'][0]['dataset']['params']['data_folder'] == f'{home}/.cache/torchok/cifar10/data'
    assert cfg.callbacks[0].params.dirpath.startswith(f'{home}/.cache/torchok/cifar10/logs/resnet18/20')
    assert cfg.task.compute_loss_on_valid is True and cfg.task.load_checkpoint is None   # schema defaults
    assert cfg.optimization[0].scheduler.pl_params.interval == 'epoch'
    assert cfg.joint_loss.normalize_weights is True and cfg.joint_loss.losses[0].tag is None
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)
    assert type(task).__name__ == 'ClassificationTask'
    assert task.input_tensors_0.shape == (1, 3, 32, 32) and str(task.input_tensors_0.dtype) == 'torch.float16'
    osl = task.configure_optimizers()
    assert type(osl[0]['optimizer']).__name__ == 'Adam' and len(osl[0]['optimizer'].param_groups) == 1
    assert type(osl[0]['lr_scheduler']['scheduler']).__name__ == 'ExponentialLR'
    assert osl[0]['lr_scheduler']['interval'] == 'epoch' and osl[0]['lr_scheduler']['monitor'] == 'val_loss'
    n_params = sum(p.numel() for p in task.parameters())
    assert n_params == 11_181_642           # ResNet-18 with a 10-way fc


def test_unknown_top_level_key_is_an_error(tmp_path):
    p = tmp_path / 'cfg.yaml'
    p.write_text(CIFAR_LIKE + '\nnonsense: 1\n')
    with pytest.raises(KeyError):
        load_config(str(p))


def test_state_dict_names_match_timm_layout():
    m = T.BACKBONES.get('resnet50')(pretrained=False)
    keys = set(m.state_dict())
    for k in ['conv1.weight', 'bn1.running_mean', 'bn1.num_batches_tracked', 'layer1.0.conv3.weight',
              'layer1.0.downsample.0.weight', 'layer1.0.downsample.1.running_var', 'layer4.2.bn3.bias']:
        assert k in keys, k
    assert sum(p.numel() for p in m.parameters()) == 23_508_032
    assert m.conv1.weight.shape == (64, 3, 7, 7)          # logical OIHW kept (checkpoint interchange)
    assert m.out_channels == 2048 and m.out_encoder_channels == (64, 256, 512, 1024, 2048)
    assert all(float(b.bn3.weight.abs().sum()) == 0 for b in m.layer1)   # zero_init_last default (resnet.py:536-539)
    with pytest.raises(RuntimeError):
        T.BACKBONES.get('resnet18')(pretrained=True)
    with pytest.raises(KeyError):
        T.BACKBONES.get('resnet18_does_not_exist')


def test_paramwise_cfg_groups():
    """mmcv-style rules of reference constructor.py:163-251."""
    from torchok_amd.constructor.constructor import Constructor
    from torchok_amd.constructor.config import to_config
    m = T.BACKBONES.get('resnet18')(pretrained=False)
    opt = Constructor.create_optimizer([m], to_config({'name': 'SGD', 'params': {'lr': 0.1, 'weight_decay': 1e-2},
                                                       'paramwise_cfg': {'norm_decay_mult': 0.0,
                                                                         'custom_keys': {'layer4': {'lr_mult': 0.1}}}}))
    groups = opt.param_groups
    assert len(groups) == len(list(m.parameters()))
    names = [n for n, _ in m.named_parameters()]
    by_name = dict(zip(names, groups))
    assert by_name['bn1.weight']['weight_decay'] == 0.0 and by_name['conv1.weight']['weight_decay'] == 1e-2
    assert abs(by_name['layer4.0.conv1.weight']['lr'] - 0.01) < 1e-12 and by_name['layer3.0.conv1.weight']['lr'] == 0.1


@pytest.mark.skipif(not os.path.exists('/root/reference/examples/configs/classification_cifar10.yaml'),
                    reason='the reference checkout exists only in the build container')
def test_reference_cifar10_recipe_runs_a_step(fake_backend):
    """BASELINE.json configs[0]: the reference's own examples/configs/classification_cifar10.yaml (ResNet-18,
    ClassificationTask, CrossEntropyLoss, Adam, Accuracy + F1Score) loads unchanged and runs a training step, an optimizer
    step and the epoch-end metric summary (datasets / trainer sections are parsed and ignored: no data pipeline here)."""
    import torch
    import torchok_amd as T
    os.environ.setdefault('HOME', '/root')
    # the recipe asks for pretrained weights (a download): overridden the way the launcher overrides keys
    cfg = T.load_config('/root/reference/examples/configs/classification_cifar10.yaml',
                        overrides={'task.params.backbone_params.pretrained': False})
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    assert type(task.backbone).__name__ == 'ResNet' and len(task.metrics_manager.phase2metrics['TRAIN']) >= 1
    opt = task.configure_optimizers()[0]['optimizer']
    assert type(opt).__name__ == 'Adam'
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 10, (4,))
    out = task.training_step({'image': x, 'target': y, 'index': torch.arange(4)}, 0)
    opt.zero_grad()
    out['loss'].backward()
    opt.step()
    task.on_train_epoch_end()
    assert any(k.startswith('train/') for k in task.logged)


def _val(t):        # every class twice: the retrieval meters need a relevant neighbour per query
    return {'image': t.randn(6, 3, 64, 64), 'target': t.tensor([0, 1, 2, 0, 1, 2])}


_RECIPES = [
    # (yaml, extra overrides, training batch, validation batch or None)
    ('classification_imagenet', {}, lambda t: {'image': t.randn(4, 3, 64, 64), 'target': t.randint(0, 1000, (4,))}, None),
    ('classification_cifar10_multi_validation', {},
     lambda t: {'image': t.randn(4, 3, 32, 32), 'target': t.randint(0, 10, (4,))}, None),
    ('segmentation_sweet_pepper', {}, lambda t: {'image': t.randn(2, 3, 64, 64), 'target': t.randint(0, 3, (2, 64, 64))}, None),
    ('pairwise_sop', {}, lambda t: {'image': t.randn(6, 3, 64, 64), 'target': t.randint(0, 3, (6,))}, _val),
    ('triplet_sop', {}, lambda t: {k: t.randn(4, 3, 64, 64) for k in ('anchor', 'positive', 'negative')}, _val),
    # semnasnet_100 is not one of the backbones of the hot path: the recipe is driven with resnet18 instead
    ('representation_arcface_sop', {'task.params.backbone_name': 'resnet18'},
     lambda t: {'image': t.randn(6, 3, 64, 64), 'target': t.randint(0, 11318, (6,))}, _val),
]


@pytest.mark.skipif(not os.path.exists('/root/reference/examples/configs'), reason='reference checkout: build container only')
@pytest.mark.parametrize('name,extra,make_batch,make_val', _RECIPES, ids=[r[0] for r in _RECIPES])
def test_shipped_recipes_drive_a_training_step(fake_backend, name, extra, make_batch, make_val):
    """The reference's own examples/configs/*.yaml (task, losses, optimizer, scheduler, metrics sections) build the task and
    run step + backward + optimizer + scheduler + epoch-end metrics without edits (only `pretrained: true` -> false);
    the metric-learning recipes also run a validation step and their HitAtKMeter at validation epoch end."""
    import torch
    os.environ.setdefault('HOME', '/root')
    cfg = T.load_config(f'/root/reference/examples/configs/{name}.yaml',
                        overrides=dict({'task.params.backbone_params.pretrained': False}, **extra))
    task = T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params).train()
    conf = task.configure_optimizers()[0]
    opt = conf['optimizer']
    torch.manual_seed(0)
    batch = make_batch(torch)
    out = task.training_step(batch, 0)
    assert torch.isfinite(out['loss'])
    opt.zero_grad()
    out['loss'].backward()
    assert sum(p.grad is not None for p in task.parameters()) > 10
    opt.step()
    if 'lr_scheduler' in conf and type(conf['lr_scheduler']['scheduler']).__name__ != 'ReduceLROnPlateau':
        conf['lr_scheduler']['scheduler'].step()
    task.on_train_epoch_end()
    if make_val is not None:
        task.eval()
        with torch.no_grad():
            task.validation_step(make_val(torch), 0)
        task.on_validation_epoch_end()
        hit = [v for k, v in task.logged.items() if 'HitAtKMeter' in k]
        assert len(hit) == 1 and 0.0 <= float(hit[0]) <= 1.0, task.logged


# ---- the fit loop (torchok_amd/run.py): what `python -m torchok` + create_trainer + Lightning's fit loop do for the hot path ----
def test_trainer_precision_is_honoured_or_refused():
    """config_structure.py:141: precision defaults to 32.  This build computes in bf16: bf16 is accepted, 16 runs as bf16 with
    a note, 32 / 64 (explicit or by default) raise instead of silently changing the recipe's arithmetic."""
    from torchok_amd.run import resolve_precision, resolve_strategy
    assert resolve_precision({'precision': 'bf16'}) == 'bf16' and resolve_precision({'precision': 'bf16-mixed'}) == 'bf16'
    assert resolve_precision({'precision': 16}) == 'bf16' and resolve_precision({'precision': '16-mixed'}) == 'bf16'
    for bad in (32, '32', 64, '64-true'):
        with pytest.raises(ValueError, match='precision'):
            resolve_precision({'precision': bad})
    with pytest.raises(ValueError, match='precision'):
        resolve_precision({})                                    # the schema default is 32
    assert resolve_strategy({'strategy': 'ddp', 'devices': 4})[:2] == (True, 4)      # classification_imagenet.yaml:121-122
    assert resolve_strategy({})[:2] == (False, 1)
    with pytest.raises(ValueError):
        resolve_strategy({'strategy': 'fsdp'})
    with pytest.raises(NotImplementedError):
        resolve_strategy({'sync_batchnorm': True})


@pytest.mark.skipif(not os.path.exists('/root/reference/examples/configs'), reason='reference checkout: build container only')
@pytest.mark.parametrize('name,extra,make_batch,make_val', _RECIPES, ids=[r[0] for r in _RECIPES])
def test_shipped_recipes_run_through_the_fit_loop(fake_backend, name, extra, make_batch, make_val):
    """Every shipped recipe, unchanged except `pretrained` and `trainer.precision=bf16`, through run.fit for 2 steps
    (task -> optimizer / scheduler -> steps -> epoch-end hooks); with its own precision (32 by default) it is refused."""
    import torch
    from torchok_amd.run import fit
    os.environ.setdefault('HOME', '/root')
    ov = dict({'task.params.backbone_params.pretrained': False}, **extra)
    asked = T.load_config(f'/root/reference/examples/configs/{name}.yaml', overrides=ov)
    wants = str((asked.trainer or {}).get('precision', 32))
    if wants in ('32', '64'):
        with pytest.raises(ValueError, match='precision'):
            fit(asked, batches=[], max_steps=0, device='cpu')
    cfg = T.load_config(f'/root/reference/examples/configs/{name}.yaml',
                        overrides=dict(ov, **{'trainer.precision': 'bf16', 'trainer.devices': 1}))
    torch.manual_seed(0)
    seen = []
    res = fit(cfg, batches=[make_batch(torch) for _ in range(2)], max_steps=2, device='cpu',
              on_step=lambda i, out: seen.append(float(out['loss'])))
    assert res['steps'] == 2 and len(seen) == 2 and all(x == x for x in seen)      # finite
    assert res['world'] == 1 and res['ranks_in_sync'] is None
    assert any(k.startswith('train/') for k in res['logged'])
    assert sum(p.grad is not None for p in res['task'].parameters()) > 10
