"""Launch-thread cost per training step with 1 process vs 8 processes on ONE box.

The first 8-GPU run will put eight ranks x (launch thread + autograd thread + RCCL proxy) on one host.  A 1-GPU box cannot run
eight RCCL ranks, but it can run eight processes that each drive the REAL launch path (engine tape, ctypes calls, allocator,
autograd thread) against `cuda:0`: the GPU is then 8x oversubscribed and irrelevant — what is measured is the HOST time a step's
launches take (`perf_counter` around `train_step`, no synchronisation inside the window; the AQL queues are deep enough for the
few steps of a window) — at a small batch, with and without the per-rank pinning bench.py applies for N > 1
(`_pin_rank`: disjoint core slice, one torch thread).

    python tools/ubench/host_contention.py            # parent: 1 and 8 workers, pinned and unpinned
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def worker(rank: int, world: int, pin: bool, backbone: str, batch: int, start_at: float):
    import torch
    import bench
    info = {}
    if pin:
        os.environ.setdefault('OMP_NUM_THREADS', '1')
        info = bench._pin_rank(rank, world)
    from torchok_amd.engine.step import train_step
    torch.cuda.set_device(0)
    torch.manual_seed(rank)
    if backbone == 'hrnet_w48':
        task = bench.build_seg_task('hrnet_w48', 19, 512, 1024).cuda().train()
        x = torch.randn(batch, 3, 512, 1024, device='cuda').to(torch.bfloat16)
        y = torch.randint(0, 19, (batch, 512, 1024), device='cuda')
    else:
        task = bench.build_task(backbone, 1000).cuda().train()
        x = torch.randn(batch, 3, 224, 224, device='cuda').to(torch.bfloat16)
        y = torch.randint(0, 1000, (batch,), device='cuda')
    opt = task.configure_optimizers()[0]['optimizer']
    b = {'image': x, 'target': y}
    for i in range(4):
        train_step(task, opt, b, i)
    torch.cuda.synchronize()
    while time.time() < start_at:          # all workers enter the measured windows together
        time.sleep(0.001)
    host, wall = [], []
    for w in range(4):                      # windows of 3 steps: enqueue time, then drain
        t0 = time.perf_counter()
        for i in range(3):
            train_step(task, opt, b, 10 + w * 3 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) / 3 * 1e3)
        wall.append((t2 - t0) / 3 * 1e3)
    print(json.dumps({'rank': rank, 'host_ms_per_step': round(min(host), 3), 'host_ms_median': round(sorted(host)[len(host) // 2], 3),
                      'wall_ms_per_step': round(min(wall), 3), 'pin': info}), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == '--worker':
        rank, world, pin, backbone, batch, start_at = sys.argv[2:8]
        worker(int(rank), int(world), pin == '1', backbone, int(batch), float(start_at))
        return
    cores = len(os.sched_getaffinity(0))
    print(f'# host: {cores} usable cores (sched_getaffinity); ResNet-50 B=16 and HRNet-W48 B=1 on cuda:0; host ms per step = time the '
          f"launch thread needs to enqueue one train_step (min over 4 windows of 3 steps)")
    for backbone, batch in (('resnet50', 16), ('hrnet_w48', 1)):
        for world in (1, 8):
            for pin in (0, 1):
                if world == 1 and pin:
                    continue
                start_at = time.time() + (45 if backbone == 'resnet50' else 75)
                procs = [subprocess.Popen([sys.executable, __file__, '--worker', str(r), str(world), str(pin), backbone, str(batch),
                                           str(start_at)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                                          env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0'))
                         for r in range(world)]
                rows = []
                for p in procs:
                    out, _ = p.communicate(timeout=600)
                    for line in out.splitlines():
                        if line.startswith('{'):
                            rows.append(json.loads(line))
                if not rows:
                    print(f'{backbone} B={batch} processes={world} pinned={pin}: no result')
                    continue
                h = [r['host_ms_per_step'] for r in rows]
                w = [r['wall_ms_per_step'] for r in rows]
                print(f'{backbone} B={batch} processes={world} pinned={pin}: host ms/step min {min(h):.2f} mean {sum(h) / len(h):.2f} '
                      f'max {max(h):.2f}   (wall incl. the shared GPU {sum(w) / len(w):.1f} ms)   {rows[0]["pin"].get("cpu_affinity", "")}',
                      flush=True)


if __name__ == '__main__':
    main()
