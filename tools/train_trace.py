"""Device timeline of Trainer.step from a rocprofv3 kernel trace: busy time per step, and the idle gaps of the GPU with
the kernels on either side.  usage (GPU box):
  cd /tmp && rocprofv3 --kernel-trace -d $OUT --output-format csv -- python3 $ROOT/tools/train_trace.py run
  python tools/train_trace.py report $OUT/*/*kernel_trace.csv"""
import csv, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
N = 12
if sys.argv[1] == 'run':
    import torch
    from fgn_amd.config import fgn_r50_c4_config
    from fgn_amd.detector import FGN
    from fgn_amd.episodes import CONFIGS, make_batch
    from fgn_amd.train import Trainer
    from fgn_amd.weights import init_state_dict
    cfg = fgn_r50_c4_config(3, 3)
    m = FGN(3, 3, state_dict=init_state_dict(cfg, 0))
    bs = [make_batch(i, 1, **CONFIGS['cfg3']) for i in range(4)]
    tr = Trainer(m)
    for i in range(4):
        tr.step(bs[i % 4])
    torch.cuda.synchronize()
    torch.zeros(1 << 20, device='cuda').fill_(1.0)        # marker kernel? (the report finds the last N steps by time)
    torch.cuda.synchronize()
    import time
    t = time.perf_counter()
    for i in range(N):
        tr.step(bs[i % 4])
    torch.cuda.synchronize()
    print('wall ms per step', (time.perf_counter() - t) / N * 1e3)
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
    # the timed region: the last N steps = from the end of the (N + 1)-th last Adagrad launch to the end of the last one
    ada = [i for i, (_, _, n) in enumerate(ev) if n.startswith('adagrad_multi_kernel')]
    k0 = ada[-(N + 1)] + 1
    ev = ev[:ada[-1] + 1]
    t_end = ev[-1][1]
    reg = ev[k0:]
    t0 = reg[0][0]
    wall = (t_end - t0) / 1e6
    # union of busy intervals
    busy, cur_s, cur_e = 0, reg[0][0], reg[0][1]
    idle = []
    last_name = reg[0][2]
    for s, e, n in reg[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            idle.append((s - cur_e, last_name, n))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
        if e >= cur_e:
            last_name = n
    busy += cur_e - cur_s
    ksum = sum(e - s for s, e, _ in reg)
    print(f'kernels {len(reg)} ({len(reg) / N:.0f} per step), wall {wall / N:.2f} ms per step, GPU busy {busy / 1e6 / N:.2f} ms per step, '
          f'sum of kernel durations {ksum / 1e6 / N:.2f} ms per step')
    idle.sort(reverse=True)
    tot_idle = sum(g for g, _, _ in idle)
    print(f'idle {tot_idle / 1e6 / N:.2f} ms per step in {len(idle) / N:.0f} gaps; gaps > 20 us: {sum(g for g, _, _ in idle if g > 20000) / 1e6 / N:.2f} ms')
    agg = {}
    for g, a, b in idle:
        k = (a[:60], b[:60])
        agg[k] = agg.get(k, [0, 0]); agg[k][0] += g; agg[k][1] += 1
    for (a, b), (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:25]:
        print(f'{g / 1e3 / N:8.1f} us/step in {c / N:5.1f} gaps  after {a}  before {b}')
