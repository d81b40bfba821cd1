#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): images/sec of the ResNet-50 ClassificationTask training step
(forward, loss, backward, SGD step, + RCCL gradient all-reduce for N > 1) at 224x224, bf16,
batch 256 per GPU, synthetic data resident in HBM.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `value` = images all ranks processed / max-over-ranks wall time of
exactly K steps (barrier + device sync on both sides).  `roofline`: the step is HBM-bound
(SURVEY.md §8(d)): achieved = algorithmic bytes of one step (309.7 MB/img x per-GPU batch) / the
average step duration measured with HIP events on the launch stream; peak = 8.0 TB/s.
`cpu_baseline`: the CPU oracle (oracle/torchok_ref.py, fp32, same step) timed on this box's host
cores on a bounded sample (rank 0, N = 1 only).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

ALG_BYTES_PER_IMG = {'resnet50': 309.7e6, 'resnet18': 70.6e6}   # SURVEY.md App. C (Model F + weights @B=256)
ALG_BYTES_SEG = {('hrnet_w48', 512, 1024): 8.08e9}               # HRNet-W48 seg 512x1024
# BASELINE.json configs[4] at its per-rank batch of 128 (SURVEY.md App. C byte model): activations 306.8 MB/img (Model F,
# independent of the batch) + 30 bytes per parameter and step / 128 images + the head's per-image rows.
#   ArcFace recipe: ResNet-50 trunk 23.51 M + PoolingLinear 2048x512 1.05 M + ArcFaceHead 11318x512 5.79 M = 30.35 M
#     parameters -> 7.11 MB/img; cosine / logit rows 11318 x (2 + 2 + 4) B = 0.09 MB/img.
#   Contrastive recipe: trunk + LinearHead 2048x512 = 24.56 M parameters -> 5.76 MB/img; the 128 x 128 distance matrix is nothing.
ALG_BYTES_C5 = {'arcface': 306.8e6 + 30 * 30.35e6 / 128 + 11318 * 8, 'contrastive': 306.8e6 + 30 * 24.56e6 / 128}
HBM_PEAK = 8.0e12
# PMC-measured HBM bytes of ONE step (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this same command,
# summarised by tools/pmc_traffic.py; corrections per MI355X_MICROARCH.md).  One file per workload, newest round first.
PMC_FILES = {
    ('resnet50', 224, 224, 256): ['r06_resnet50_bs256_pmc_traffic.json', 'r05_resnet50_bs256_pmc_traffic.json', 'r04_resnet50_bs256_pmc_traffic.json', 'r03_resnet50_bs256_pmc_traffic.json', 'r02_resnet50_bs256_pmc_traffic.json',
                                 'r01_resnet50_bs256_pmc_traffic.json'],
    ('swinv2_custom', 224, 224, 256): ['r06_swinv2t_224_bs256_pmc_traffic.json', 'r05_swinv2t_224_bs256_pmc_traffic.json', 'r04_swinv2t_224_bs256_pmc_traffic.json', 'r03_swinv2t_224_bs256_pmc_traffic.json', 'r02_swinv2t_224_bs256_pmc_traffic.json'],
    ('davit_t', 224, 224, 256): ['r02_davit_t_224_bs256_pmc_traffic.json'],
    ('hrnet_w48', 512, 1024, 24): ['r06_hrnet_w48_512x1024_bs24_pmc_traffic.json', 'r05_hrnet_w48_512x1024_bs24_pmc_traffic.json', 'r04_hrnet_w48_512x1024_bs24_pmc_traffic.json', 'r03_hrnet_w48_512x1024_bs24_pmc_traffic.json', 'r02_hrnet_w48_512x1024_bs24_pmc_traffic.json'],
    ('hrnet_w48', 512, 1024, 8): ['r02_hrnet_w48_512x1024_bs8_pmc_traffic.json'],
}


def measured_traffic(backbone: str, res: int, width: int, batch: int):
    """(GB per step, file) of the committed PMC measurement of this exact workload, or (None, None)."""
    for name in PMC_FILES.get((backbone, res, width, batch), []):
        path = os.path.join(ROOT, 'profiles', name)
        if os.path.exists(path):
            with open(path) as f:
                return round(json.load(f)['hbm_bytes_per_step'] / 1e9, 2), 'profiles/' + name
    return None, None


def traffic_note(traffic_file):
    """`roofline.traffic` is NOT measured by the run that prints it: rocprofv3 collects PMC counters from outside the process
    (separate FETCH_SIZE / WRITE_SIZE passes, tools/profile_workload.sh).  The line carries the committed figure of the same
    command on the build it was profiled on and says so."""
    if not traffic_file:
        return None
    return (f'committed profile {traffic_file} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, FETCH x2 per '
            'MI355X_MICROARCH.md, calibrated on the optimizer kernel); a constant of the profiled build, not measured in this run')


def measured_mfma_util(backbone: str, res: int, width: int, batch: int):
    """Counter-based MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES over GRBM_GUI_ACTIVE x 1024 SIMDs, tools/pmc_mfma.py) of the
    committed PMC pass of this exact workload, or None."""
    for name in PMC_FILES.get((backbone, res, width, batch), []):
        path = os.path.join(ROOT, 'profiles', name.replace('pmc_traffic', 'pmc_mfma'))
        if os.path.exists(path):
            with open(path) as f:
                return round(json.load(f)['mfma_util_busy_over_active'], 4)
    return None


def build_seg_task(backbone: str, num_classes: int, h: int, w: int):
    """HRNet + HRNetSegmentationNeck + SegmentationHead + CrossEntropyLoss (secondary workload,
    `--backbone hrnet_w48 --res 512 --width 1024 --classes 19 --batch 8`; never the default line)."""
    import torchok_amd as T
    from torchok_amd.constructor.config import apply_schema
    cfg = apply_schema({
        'task': {'name': 'SegmentationTask',
                 'params': {'backbone_name': backbone, 'backbone_params': {'pretrained': False, 'in_channels': 3},
                            'neck_name': 'HRNetSegmentationNeck', 'head_name': 'SegmentationHead',
                            'head_params': {'num_classes': num_classes},
                            'inputs': [{'shape': [3, h, w], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': 'CrossEntropyLoss', 'params': {'ignore_index': 255},
                                   'mapping': {'input': 'prediction', 'target': 'target'}}]},
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.01, 'weight_decay': 5e-4, 'momentum': 0.9}}}],
        'data': {}, 'trainer': {'precision': 'bf16', 'strategy': 'ddp'},
    })
    return T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)


ALG_FLOPS_PER_IMG = {('swinv2_custom', 224): 26.94e9,            # SwinV2-T 224 (window 7)
                     # DaViT-T 224: 4.496 GMAC forward (linears 12*T*C^2 per block, window 98*T*C + channel 64*T*C attention,
                     # patch embeds, head) x 2 x 3 (forward + dgrad + wgrad)
                     ('davit_t', 224): 26.98e9}
MFMA_PEAK_BF16 = 2.5e15


def build_swin_task(num_classes: int, res: int, backbone: str = 'swinv2_custom'):
    """SwinV2-T (embed 96, depths 2-2-6-2, heads 3-6-12-24, window 7) + ClassificationTask + AdamW
    (secondary workload: `--backbone swinv2_custom --res 224 --batch 128`; never the default line).  `davit_t`: the
    reference's in-tree DaViT-T under the same task / optimizer."""
    import torchok_amd as T
    from torchok_amd.constructor.config import apply_schema
    cfg = apply_schema({
        'task': {'name': 'ClassificationTask',
                 'params': {'backbone_name': backbone,
                            'backbone_params': {'pretrained': False, 'in_channels': 3, 'img_size': res, 'window_size': 7,
                                                'drop_path_rate': 0.1},
                            'pooling_name': 'Pooling', 'head_name': 'ClassificationHead',
                            'head_params': {'num_classes': num_classes},
                            'inputs': [{'shape': [3, res, res], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction', 'target': 'target'}}]},
        'optimization': [{'optimizer': {'name': 'AdamW', 'params': {'lr': 1e-3, 'weight_decay': 0.05}}}],
        'data': {}, 'trainer': {'precision': 'bf16', 'strategy': 'ddp'},
    })
    return T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)


def build_task(backbone: str, num_classes: int):
    import torchok_amd as T
    from torchok_amd.constructor.config import apply_schema
    cfg = apply_schema({
        'task': {'name': 'ClassificationTask',
                 'params': {'backbone_name': backbone,
                            'backbone_params': {'pretrained': False, 'in_channels': 3, 'zero_init_last': False},
                            'pooling_name': 'Pooling', 'head_name': 'ClassificationHead',
                            'head_params': {'num_classes': num_classes},
                            'inputs': [{'shape': [3, 224, 224], 'dtype': 'float32'}]}},
        'joint_loss': {'losses': [{'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction', 'target': 'target'}}]},
        # examples/configs/classification_imagenet.yaml:29-35
        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.1, 'weight_decay': 1e-4, 'momentum': 0.9}}}],
        'data': {}, 'trainer': {'precision': 'bf16', 'strategy': 'ddp'},
    })
    return T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)


def build_c5_task(kind: str):
    """BASELINE.json configs[4] at the per-rank shape (bs 1024 over 8 GPUs = 128 / GPU, 3x224x224): ResNet-50 +
    PoolingLinear(512) + ArcFaceHead(11318) + CrossEntropyLoss (`representation_arcface_sop.yaml:1-24`), or
    ResNet-50 + Pooling + LinearHead(512, normalize) + ContrastiveLoss under PairwiseLearnTask (`pairwise_sop.yaml`)."""
    import torchok_amd as T
    from torchok_amd.constructor.config import apply_schema
    bb = {'backbone_name': 'resnet50', 'backbone_params': {'pretrained': False, 'in_channels': 3},
          'inputs': [{'shape': [3, 224, 224], 'dtype': 'float32'}]}
    if kind == 'arcface':
        task = {'name': 'ClassificationTask',
                'params': dict(bb, pooling_name='PoolingLinear', pooling_params={'out_channels': 512},
                               head_name='ArcFaceHead', head_params={'num_classes': 11318})}
        loss = {'name': 'CrossEntropyLoss', 'mapping': {'input': 'prediction', 'target': 'target'}}
    else:
        task = {'name': 'PairwiseLearnTask',
                'params': dict(bb, pooling_name='Pooling', head_name='LinearHead',
                               head_params={'out_channels': 512, 'normalize': True}, num_classes=11318)}
        loss = {'name': 'ContrastiveLoss', 'params': {'margin': 0.5}, 'mapping': {'emb1': 'emb1', 'emb2': 'emb2', 'R': 'R'}}
    cfg = apply_schema({'task': task, 'joint_loss': {'losses': [loss]},
                        'optimization': [{'optimizer': {'name': 'SGD', 'params': {'lr': 0.01, 'momentum': 0.9}}}],
                        'data': {}, 'trainer': {'precision': 'bf16', 'strategy': 'ddp'}})
    return T.TASKS.get(cfg.task.name)(cfg, **cfg.task.params)


def secondary_block(budget_s: float = 20.0, warmup: int = 3, steps: int = 10):
    """The other BASELINE.json configs at their per-GPU shapes, 3 + 10 steps each (bounded: a workload is skipped once the
    budget is spent), so that the driver's record carries them next to the headline: {ms_per_step, frac, bound}."""
    from torchok_amd.engine.step import train_step
    plans = [
        ('swinv2_t_224_bs256_adamw', lambda: build_swin_task(1000, 224), 256, (3, 224, 224), 1000, False,
         ('mfma', ALG_FLOPS_PER_IMG[('swinv2_custom', 224)], MFMA_PEAK_BF16),
         'BASELINE config 3: SwinV2-T (embed 96, depths 2-2-6-2, window 7, drop_path 0.1) + ClassificationTask(Pooling, '
         'ClassificationHead 1000) + CrossEntropyLoss + AdamW, synthetic 3x224x224 bf16, batch 256 on one GPU (the per-rank '
         'shape of its DDP runs)'),
        ('hrnet_w48_seg_512x1024_bs24', lambda: build_seg_task('hrnet_w48', 19, 512, 1024), 24, (3, 512, 1024), 19, True,
         ('hbm', ALG_BYTES_SEG[('hrnet_w48', 512, 1024)], HBM_PEAK),
         'BASELINE config 4: HRNet-W48 + HRNetSegmentationNeck + SegmentationHead(19) + SegmentationTask + CrossEntropyLoss + '
         'SGD, synthetic 3x512x1024 bf16 (Cityscapes shape), batch 24 on one GPU (the per-rank shape of the 8-GPU run)'),
        ('resnet50_arcface11318_bs128', lambda: build_c5_task('arcface'), 128, (3, 224, 224), 11318, False,
         ('hbm', ALG_BYTES_C5['arcface'], HBM_PEAK),
         'BASELINE config 5 at its PER-RANK shape (global batch 1024 over 8 GPUs = 128 per rank; pure data parallelism, so the '
         'rank sees exactly this): ResNet-50 + PoolingLinear(512) + ArcFaceHead(11318) + CrossEntropyLoss + SGD, 3x224x224 bf16'),
        ('resnet50_contrastive_bs128', lambda: build_c5_task('contrastive'), 128, (3, 224, 224), 11318, False,
         ('hbm', ALG_BYTES_C5['contrastive'], HBM_PEAK),
         'config 5 recipe variant at its per-rank shape (128 of 1024): ResNet-50 + PoolingLinear + LinearHead + ContrastiveLoss '
         '(PairwiseLearnTask) + SGD, 3x224x224 bf16'),
    ]
    out, t_start = {}, time.perf_counter()
    for name, build, bsz, shape, classes, seg, (bound, per_img, peak), workload in plans:
        if time.perf_counter() - t_start > budget_s:
            out[name] = {'skipped': f'secondary budget of {budget_s:.0f} s spent'}
            continue
        try:
            task = build().cuda().train()
            opt = task.configure_optimizers()[0]['optimizer']
            g = torch.Generator(device='cuda').manual_seed(4321)
            batch = {'image': torch.randn(bsz, *shape, generator=g, device='cuda').to(torch.bfloat16),
                     'target': torch.randint(0, classes, (bsz, *shape[1:]) if seg else (bsz,), generator=g, device='cuda')}
            for i in range(warmup):
                train_step(task, opt, batch, i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(steps):
                out_ = train_step(task, opt, batch, warmup + i)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / steps
            ach = per_img * bsz / (ms * 1e-3)
            out[name] = {'workload': workload, 'ms_per_step': round(ms, 3), 'images_per_sec': round(bsz / ms * 1e3, 1),
                         'bound': bound,
                         'frac': round(ach / peak, 4),
                         'algorithmic_per_img': f'{per_img / 1e9:.2f} GFLOP' if bound == 'mfma' else f'{per_img / 1e6:.1f} MB',
                         'batch': bsz, 'steps': steps, 'warmup': warmup,
                         'loss_finite': bool(torch.isfinite(out_['loss']))}
            del task, opt, batch, out_
        except Exception as e:      # a secondary workload must never take the headline line down with it
            out[name] = {'error': f'{type(e).__name__}: {e}'[:200]}
        import gc
        gc.collect()
        torch.cuda.empty_cache()
    out['wall_s'] = round(time.perf_counter() - t_start, 1)
    return out


def _usable_cores() -> int:
    """Cores this process may really use: affinity mask, capped by the cgroup CPU quota (a box can
    show 256 logical CPUs under a much smaller quota; oversubscribing them makes torch crawl)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def _pin_rank(local_rank: int, local_world: int):
    """N > 1: one launch thread per rank.  Each rank's process keeps to a disjoint slice of the cores it was given
    (`os.sched_setaffinity`) and runs torch's host-side ops on ONE thread: eight ranks x (launch thread + autograd thread +
    RCCL proxy) on a host whose cgroup may hand the job 16 cores must not also start eight intra-op pools of 16 threads
    each (profiles/r05_host_contention.txt).  Returns a short description for the bench line."""
    torch.set_num_threads(1)
    try:
        torch.set_num_interop_threads(1)
    except RuntimeError:
        pass                                    # already started: harmless
    if not hasattr(os, 'sched_getaffinity') or local_world <= 1:
        return {'host_threads_per_rank': 1}
    cores = sorted(os.sched_getaffinity(0))
    per = max(1, len(cores) // local_world)
    mine = cores[local_rank * per:(local_rank + 1) * per] or cores
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return {'host_threads_per_rank': 1}
    return {'host_threads_per_rank': 1, 'cpu_affinity': f'{len(mine)} of {len(cores)} cores ({mine[0]}..{mine[-1]})'}


def _cpu_model() -> str:
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except Exception:
        pass
    import platform
    return platform.processor() or platform.machine()


def cpu_baseline(backbone: str, num_classes: int, res: int, width: int = 0, batch: int = 0, steps: int = 5,
                 warmup: int = 2, budget_s: float = 45.0):
    """The oracle's training step (fp32, the reference's `trainer.accelerator='cpu'` arithmetic) on the host cores:
    a bounded sample — `warmup` + `steps` steps at a batch that fits the host, cut short when a step is so slow
    that the default run would not finish within minutes.  A reported baseline, not the target."""
    threads = _usable_cores()
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    width = width or res
    if backbone.startswith('hrnet'):
        import oracle.hrnet_ref as H
        batch = batch or 1
        ref = H.SegmentationModel(backbone, num_classes).train()
        opt = torch.optim.SGD(ref.parameters(), lr=0.01, momentum=0.9, weight_decay=5e-4)
        x, y = torch.randn(batch, 3, res, width), torch.randint(0, num_classes, (batch, res, width))
        ce = torch.nn.CrossEntropyLoss(ignore_index=255)

        def one():
            opt.zero_grad(set_to_none=True)
            ce(ref.forward_with_gt({'image': x, 'target': y})['prediction'], y).backward()
            opt.step()
    elif backbone in ('swinv2_custom', 'davit_t'):
        batch = batch or 16
        if backbone == 'swinv2_custom':
            import oracle.swin_ref as S
            bb = S.SwinV2(img_size=res, window_size=7, drop_path_rate=0.1)
        else:
            import oracle.davit_ref as D
            bb = D.davit_t(drop_path_rate=0.1)
        ref = torch.nn.Sequential(bb, torch.nn.AdaptiveAvgPool2d(1), torch.nn.Flatten(1),
                                  torch.nn.Linear(bb.out_channels, num_classes)).train()
        opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=0.05)
        x, y = torch.randn(batch, 3, res, res), torch.randint(0, num_classes, (batch,))

        def one():
            opt.zero_grad(set_to_none=True)
            torch.nn.functional.cross_entropy(ref(x), y).backward()
            opt.step()
    else:
        import oracle.torchok_ref as R
        batch = batch or 16
        ref = R.ClassificationModel(backbone, num_classes, zero_init_last=False).train()
        opt = torch.optim.SGD(ref.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        x, y = torch.randn(batch, 3, res, res), torch.randint(0, num_classes, (batch,))

        def one():
            R.training_step(ref, {'image': x, 'target': y}, opt)
    t_all = time.perf_counter()
    done_w = 0
    for _ in range(warmup):
        one()
        done_w += 1
        if time.perf_counter() - t_all > budget_s * 0.3:
            break
    per = (time.perf_counter() - t_all) / done_w
    steps = max(1, min(steps, int((budget_s - (time.perf_counter() - t_all)) / max(per, 1e-3))))
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    dt = time.perf_counter() - t0
    return {'value': batch * steps / dt, 'unit': 'images/sec', 'cores': threads, 'cpu_model': _cpu_model(), 'kind': 'port',
            'sample': f'{steps} fp32 steps ({done_w} warm-up) of the same {backbone} {res}x{width} training step at batch '
                      f'{batch}, oracle/ on torch {torch.__version__} CPU, {threads} threads'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=256, help='per-GPU batch')
    ap.add_argument('--backbone', default='resnet50')
    ap.add_argument('--res', type=int, default=224)
    ap.add_argument('--width', type=int, default=0, help='image width when it differs from --res (segmentation)')
    ap.add_argument('--classes', type=int, default=1000)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the bounded secondary-workload block of the default line')
    ap.add_argument('--graph', type=int, default=-1, help='1: replay the step as one hipGraph, 0: eager launches '
                    '(default: graph only where the step is host-launch-bound: HRNet below batch 16)')
    args = ap.parse_args()

    # stdout carries exactly ONE line (the JSON result): native libraries that chat on file descriptor 1 (RCCL's
    # version banner at communicator creation) are sent to stderr until that line is printed
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if args.gpus != world and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (torch.cuda.is_available() is False)')
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist
    # TOK_BENCH_FORCE_DIST=1: run the N>1 code path (RCCL group, bucketed gradient all-reduce, barriers) with one
    # rank — the way to exercise it on a 1-GPU box; the numbers it prints include the reducer's overhead
    dist_on = world > 1 or os.environ.get('TOK_BENCH_FORCE_DIST') == '1'
    host_pin = None
    if world > 1:
        # before RCCL starts its proxy threads (they inherit the mask); the env defaults cover libraries that read them lazily
        os.environ.setdefault('OMP_NUM_THREADS', '1')
        os.environ.setdefault('MKL_NUM_THREADS', '1')
        host_pin = _pin_rank(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', str(world))))
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group('nccl', rank=rank, world_size=world)

    import __graft_entry__ as ge
    if not os.path.exists(ge.LIB):
        ge.build()

    torch.manual_seed(1234)
    seg = args.backbone.startswith('hrnet')
    width = args.width or args.res
    swin = args.backbone.startswith('swinv2') or args.backbone.startswith('davit')
    if swin and args.backbone not in ('swinv2_custom', 'davit_t'):
        raise SystemExit('bench.py: the transformer workloads are swinv2_custom (SwinV2-T geometry, window 7) and davit_t at --res')
    def make_task():
        t = (build_seg_task(args.backbone, args.classes, args.res, width) if seg
             else build_swin_task(args.classes, args.res, args.backbone) if swin
             else build_task(args.backbone, args.classes)).cuda().train()
        return t, t.configure_optimizers()[0]['optimizer']

    # N > 1: the single-rank step of THIS run first — a throw-away replica stepped without a reducer on
    # every rank at once (same box, same host contention, no exchange) — so that the line can state what the exchange and
    # the shared host cost against it.  Its parameters and optimizer state are discarded: the timed replica below is built
    # fresh and broadcast from rank 0.
    local_probe = None
    if dist_on:
        from torchok_amd.engine.step import train_step as _ts
        t0_, o0_ = make_task()
        g0 = torch.Generator(device='cuda').manual_seed(99 + rank)
        im0 = torch.randn(args.batch, 3, args.res, width, generator=g0, device='cuda', dtype=torch.float32).to(torch.bfloat16)
        tg0 = torch.randint(0, args.classes, (args.batch, args.res, width) if seg else (args.batch,), generator=g0, device='cuda')
        b0 = {'image': im0, 'target': tg0}
        nw, ns = min(args.warmup, 5), max(3, min(args.steps, 15))
        for i in range(nw):
            _ts(t0_, o0_, b0, i, None)
        torch.cuda.synchronize()
        dist.barrier()
        e_ = [torch.cuda.Event(enable_timing=True) for _ in range(ns + 1)]
        e_[0].record()
        for i in range(ns):
            _ts(t0_, o0_, b0, nw + i, None)
            e_[i + 1].record()
        torch.cuda.synchronize()
        local_probe = statistics.median([e_[i].elapsed_time(e_[i + 1]) for i in range(ns)])
        del t0_, o0_, b0, im0, tg0, e_
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        dist.barrier()

    task, opt = make_task()
    reducer = None
    if dist_on:
        from torchok_amd.dist import GradientAllReducer
        # (the transformer backbones keep per-stage feature norms that a classification forward never runs: torch DDP needs
        # find_unused_parameters=True for them as well, reference swin.py:129-148)
        reducer = GradientAllReducer(opt, module=task,     # + DDP's rank-0 buffer broadcast (one flat collective per dtype)
                                     find_unused_parameters=True if swin else None,
                                     # the unused set of these backbones is the same in every step: no per-step host read
                                     static_unused_pattern=True if swin else None)
        reducer.enable_timing()

    g = torch.Generator(device='cuda').manual_seed(1234 + rank)
    image = torch.randn(args.batch, 3, args.res, width, generator=g, device='cuda', dtype=torch.float32).to(torch.bfloat16)
    target = torch.randint(0, args.classes, (args.batch, args.res, width) if seg else (args.batch,), generator=g,
                           device='cuda')
    batch = {'image': image, 'target': target}

    use_graph = (args.graph == 1 or (args.graph < 0 and seg and args.batch < 16)) and world == 1
    graphed = None
    if use_graph:
        from torchok_amd.engine.graph import GraphedTrainingStep
        for g_ in opt.param_groups:          # Adam / AdamW: device-side step count (torch's capturable=True)
            if 'capturable' in g_:
                g_['capturable'] = True
        graphed = GraphedTrainingStep(task, opt, batch, reducer=reducer)

    from torchok_amd.engine.step import replicas_in_sync, train_step

    def step(i):
        # ONE function for every caller (engine/step.py): training_step -> backward (+ bucketed exchange) -> optimizer ->
        # on_train_batch_end (the per-step loss mean of reference tasks/base.py:163-173, on the comm stream for N > 1)
        if graphed is not None:
            return graphed(batch)['loss']
        return train_step(task, opt, batch, i, reducer)['loss']

    for i in range(args.warmup):
        loss = step(i)
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
        reducer.enable_timing()          # drop the warm-up steps' event pairs
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    evs[0].record()
    host_done = []
    for i in range(args.steps):
        loss = step(args.warmup + i)
        evs[i + 1].record()
        host_done.append(time.perf_counter())
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    # how far the launch thread runs ahead of the GPU: evs[0] was recorded on an idle device at host time t0, so step i's
    # kernels finish at about t0 + elapsed(evs[0], evs[i + 1]); the host finished enqueuing that step at host_done[i]
    host_lead_ms = [evs[0].elapsed_time(evs[i + 1]) - (host_done[i] - t0) * 1e3 for i in range(args.steps)]
    final_loss = float(loss.detach())
    in_sync = replicas_in_sync(reducer)      # N > 1: every rank must hold bit-identical parameters after the run
    per_rank = None
    if dist_on:
        t = torch.tensor([dt], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # one row per rank: [host lead min, host lead p50, step p50, single-rank step p50 of this run, exposed exchange p50 / max]
        ex = reducer.exposed_comm_ms() or [0.0]
        mine = torch.tensor([min(host_lead_ms), statistics.median(host_lead_ms), statistics.median(step_ms), local_probe,
                             statistics.median(ex), max(ex)], device='cuda', dtype=torch.float64)
        rows = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(rows, mine)
        per_rank = [[round(float(v), 3) for v in r.tolist()] for r in rows]

    if rank == 0:
        value = args.batch * world * args.steps / dt
        alg = ALG_BYTES_SEG.get((args.backbone, args.res, width)) if seg else \
            (ALG_BYTES_PER_IMG.get(args.backbone) if args.res == 224 else None)
        ev_ms = sum(step_ms) / len(step_ms)
        roofline = None
        traffic, traffic_file = measured_traffic(args.backbone, args.res, width, args.batch)
        flops = ALG_FLOPS_PER_IMG.get((args.backbone, args.res)) if swin else None
        if flops is not None:
            ach = flops * args.batch / (ev_ms * 1e-3) / 1e12
            roofline = {'bound': 'mfma', 'achieved': round(ach, 1), 'peak': MFMA_PEAK_BF16 / 1e12, 'unit': 'TFLOP/s',
                        'frac': round(ach * 1e12 / MFMA_PEAK_BF16, 4), 'traffic': traffic,
                        'traffic_unit': 'GB/step' if traffic_file else None, 'traffic_source': traffic_note(traffic_file),
                        'launch': f'one training step, HIP-event avg {ev_ms:.3f} ms'}
        elif alg is not None:
            achieved = alg * args.batch / (ev_ms * 1e-3) / 1e9
            roofline = {'bound': 'hbm', 'achieved': round(achieved, 1), 'peak': HBM_PEAK / 1e9, 'unit': 'GB/s',
                        'frac': round(achieved * 1e9 / HBM_PEAK, 4),
                        'traffic': traffic, 'traffic_unit': 'GB/step' if traffic_file else None,
                        'traffic_source': traffic_note(traffic_file),
                        'algorithmic': round(alg * args.batch / 1e9, 2), 'algorithmic_unit': 'GB/step',
                        'launch': 'one training step (all kernels of fwd+bwd+optimizer on the step stream), '
                                  f'HIP-event avg {ev_ms:.3f} ms'}
        line = {
            'metric': 'images/sec (node) ResNet-50 224px bs256/GPU; step p50 ms' if args.backbone == 'resnet50'
                      else f'images/sec (node) {args.backbone} {args.res}px',
            'value': round(value, 1), 'unit': 'images/sec', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3),
            'step_p50_ms': round(statistics.median(step_ms), 3),
            'host_lead_ms': {'min': round(min(host_lead_ms), 2), 'p50': round(statistics.median(host_lead_ms), 2)},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'bf16',
            'data': 'synthetic', 'final_loss': round(final_loss, 4),
            'config': {'workload': (f'{args.backbone} + SegmentationTask(HRNetSegmentationNeck, SegmentationHead '
                                    f'{args.classes}) + CrossEntropyLoss + SGD(momentum 0.9, wd 5e-4), synthetic '
                                    f'3x{args.res}x{width} bf16, batch {args.batch}/GPU') if seg else
                                   (f'{"DaViT-T" if args.backbone == "davit_t" else "SwinV2-T"}(window 7, drop_path 0.1) + ClassificationTask(Pooling, ClassificationHead '
                                    f'{args.classes}) + CrossEntropyLoss + AdamW, synthetic 3x{args.res}x{args.res} bf16, '
                                    f'batch {args.batch}/GPU') if swin else
                                   f'{args.backbone} + ClassificationTask(Pooling, ClassificationHead {args.classes}) '
                                   f'+ CrossEntropyLoss + SGD(momentum 0.9, wd 1e-4), synthetic 3x{args.res}x{args.res} '
                                   f'bf16, batch {args.batch}/GPU', 'global_batch': args.batch * world,
                       'parallelism': f'dp{world}', 'launch_mode': 'hipGraph replay' if use_graph else 'eager'},
            'roofline': roofline,
        }
        try:      # fp32 elements the optimizer kernel walks (padded arena): what tools/pmc_traffic.py calibrates the counters on
            line['config']['arena_elements'] = int(sum(a.total for a in opt._arenas if a is not None))
        except Exception:
            pass
        if roofline is not None:
            # (a constant read from the committed PMC pass of this workload, like `traffic`: counters cannot be collected
            #  from inside the timed process)
            roofline['mfma_util_committed_profile'] = measured_mfma_util(args.backbone, args.res, width, args.batch)
        if dist_on:
            # self-diagnosis of a multi-rank run: who is slow (per-rank step p50), whether a launch thread starves its GPU
            # (host_lead_ms.min <= 0 on some rank), how much of the exchange the backward did not hide, and what the step costs
            # against the single-rank step measured in this same run on the same (shared) host
            cols = list(zip(*per_rank))
            single = statistics.mean(cols[3])
            line['scaling_diagnosis'] = {
                'per_rank': {'host_lead_ms_min': list(cols[0]), 'host_lead_ms_p50': list(cols[1]), 'step_p50_ms': list(cols[2]),
                             'single_rank_step_p50_ms': list(cols[3]), 'exposed_exchange_ms_p50': list(cols[4]),
                             'exposed_exchange_ms_max': list(cols[5])},
                'slowest_rank': int(max(range(len(cols[2])), key=lambda r_: cols[2][r_])),
                'host_bound_ranks': [r_ for r_ in range(len(cols[0])) if cols[0][r_] <= 0.0],
                'single_rank_step_ms_same_run': round(single, 3),
                'weak_scaling_efficiency_same_run': round(single / (dt / args.steps * 1e3), 4),
                'note': 'single_rank_step = a throw-away replica stepped without a reducer on every rank at once before the timed '
                        'region (same box, same host contention, no exchange); exposed_exchange = step-stream time between the end '
                        'of backward and the join of every bucket / buffer broadcast (incl. cast / scale kernels); the driver '
                        'computes the official efficiency from its own N=1 run',
            }
            line['config']['rccl_ranks'] = dist.get_world_size()
            if host_pin is not None:
                line['config']['host'] = host_pin
            line['ranks_in_sync'] = in_sync
            line['config']['per_step_collectives'] = 'gradient buckets + buffer broadcast + loss mean (async, comm stream)'
            line['config']['grad_exchange'] = f"bucketed all-reduce(AVG), {'bf16' if reducer.bf16 else 'fp32'} payload, " \
                                              f"{sum(len(b) for b in reducer.buckets)} buckets + buffer broadcast"
        if world == 1 and not dist_on and not args.no_secondary and args.backbone == 'resnet50' and args.batch == 256:
            # free the headline workload first: the secondary ones need the memory, and their numbers must not depend on it
            del task, opt, batch, image, target, loss
            import gc
            gc.collect()
            torch.cuda.empty_cache()
            line['secondary'] = secondary_block()
        if world == 1 and not args.no_cpu_baseline:
            line['cpu_baseline'] = cpu_baseline(args.backbone, args.classes, args.res, width)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(line) + '\n').encode())
    if dist_on:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
