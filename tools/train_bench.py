"""Time of one training step on cfg3-size episodes (800x1333 query, 9 supports of 256^2, 63 000 anchors, 12 000 -> 2000
proposals, 128 sampled RoIs per image): forward_train alone (losses) and Trainer.step (forward + backward of the heads +
Adagrad + re-pack), beside the CPU oracle's forward_train and forward+autograd-backward.
usage (GPU box): python tools/train_bench.py [--batch 1] [--steps 10] [--out profiles/r02_train_step.json]"""
import argparse, json, os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from fgn_amd.config import fgn_r50_c4_config
from fgn_amd.detector import FGN
from fgn_amd.episodes import CONFIGS, make_batch
from fgn_amd.train import Trainer
from fgn_amd.weights import init_state_dict

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=1)
ap.add_argument('--steps', type=int, default=10)
ap.add_argument('--cpu', type=int, default=1)
ap.add_argument('--out', default='')
a = ap.parse_args()
cfg = fgn_r50_c4_config(3, 3)
sd = init_state_dict(cfg, 0)
m = FGN(3, 3, state_dict=sd)
batches = [make_batch(i * a.batch, a.batch, **CONFIGS['cfg3']) for i in range(4)]
if not os.environ.get('TRAIN_BENCH_PAGEABLE'):
    # page-locked input tensors, the protocol of bench.py (and of a DataLoader with pin_memory=True): a pageable 12.8 MB
    # image is uploaded in staging-buffer pieces with the host copying between them - ~1 ms of every step with the GPU idle
    batches = [{k: (v.pin_memory() if isinstance(v, torch.Tensor) else
                    [t.pin_memory() if isinstance(t, torch.Tensor) else t for t in v] if isinstance(v, (list, tuple)) else v)
                for k, v in b.items()} for b in batches]


def timed(fn, n):
    fn(batches[0]); fn(batches[1])
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(batches[i % 4])
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


res = {'workload': f'cfg3 training step, batch {a.batch}: 3-way 3-shot, 800x1333, ResNet-50-C4 (frozen), heads trained',
       'batch': a.batch, 'inputs': 'pageable host tensors' if os.environ.get('TRAIN_BENCH_PAGEABLE') else 'page-locked host tensors'}
res['forward_train_ms'] = timed(lambda b: m.forward_train(**b), a.steps)
tr = Trainer(m)
res['forward_backward_ms'] = timed(lambda b: tr.forward_backward(b), a.steps)
res['train_step_ms'] = timed(lambda b: tr.step(b), a.steps)
L = tr.step(batches[0])
res['losses'] = {k: (float(v[0]) if isinstance(v, list) else float(v)) for k, v in L.items()}
res['peak_mem_gb'] = torch.cuda.max_memory_allocated() / 2 ** 30
if a.cpu:
    from oracle import fgn_train_cpu as T
    s2 = {k: v.clone() for k, v in sd.items()}
    t = time.perf_counter()
    T.forward_train(s2, cfg, **batches[0])
    res['cpu_oracle_forward_train_ms'] = (time.perf_counter() - t) * 1e3
    for k, v in s2.items():
        if k.startswith(T.TRAINABLE_PREFIXES) and v.is_floating_point() and 'running_' not in k:
            v.requires_grad_(True)
    t = time.perf_counter()
    T.total_loss(T.forward_train(s2, cfg, grad=True, **batches[1])).backward()
    res['cpu_oracle_forward_backward_ms'] = (time.perf_counter() - t) * 1e3
    res['cpu_threads'] = torch.get_num_threads()
print(json.dumps(res))
if a.out:
    json.dump(res, open(os.path.join(R, a.out), 'w'), indent=1)
