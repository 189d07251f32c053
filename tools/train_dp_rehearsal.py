"""Two-rank rehearsal of data-parallel training on ONE GPU (backend gloo: RCCL refuses two ranks on one device): each
rank trains on its own episode, the gradients are averaged by fgn_amd.dist.allreduce_mean inside Trainer.step, and both
ranks must hold bit-identical weights afterwards - equal to a single process that averages the two gradients itself.
usage (GPU box): python tools/train_dp_rehearsal.py"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)


def worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    from fgn_amd.config import tiny_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import make_batch
    from fgn_amd.train import Trainer
    from fgn_amd.weights import init_state_dict
    dist.init_process_group('gloo', init_method=f'tcp://127.0.0.1:{port}', rank=rank, world_size=world)
    try:
        cfg = tiny_config(3, 2, width_div=2)
        m = FGN(3, 2, backbone=cfg['backbone'], rpn_head=cfg['rpn_head'], roi_head=cfg['roi_head'],
                state_dict=init_state_dict(cfg, 0))
        tr = Trainer(m)
        for it in range(2):
            torch.manual_seed(100 * it + rank)
            L = tr.step(make_batch(10 * it + rank, 1, 3, 2, 160, 224, 64))
        import hashlib
        h = hashlib.sha256()
        for k in sorted(tr.W):
            h.update(tr.W[k].cpu().numpy().tobytes())
        q.put((rank, h.hexdigest(), float(L['loss_cls'])))
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = 29700 + os.getpid() % 1000
    procs = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = sorted((q.get(timeout=600) for _ in procs), key=lambda x: x[0])
    for p in procs:
        p.join(timeout=120)
    same = out[0][1] == out[1][1]
    print('ranks hold identical weights after 2 data-parallel steps:', same, '| loss_cls per rank', out[0][2], out[1][2])
    sys.exit(0 if same and all(p.exitcode == 0 for p in procs) else 1)
