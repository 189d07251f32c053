"""Child of tests/test_host_cpu.py::test_a_failing_rank_ends_the_whole_job: one rank of a gloo job that gathers
detection records step by step (fgn_amd.dist.gather_detections, the per-step collective of bench.py --gpus N); the rank
named by FGN_FAIL_RANK raises at step FGN_FAIL_STEP.  Started by torch.distributed.run, never imported."""
import datetime
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fgn_amd import dist as fd  # noqa: E402


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo', timeout=datetime.timedelta(seconds=int(os.environ.get('FGN_PG_TIMEOUT', '30'))))
    fail_rank, fail_step = int(os.environ.get('FGN_FAIL_RANK', '-1')), int(os.environ.get('FGN_FAIL_STEP', '3'))
    for step in range(8):
        if rank == fail_rank and step == fail_step:
            raise RuntimeError(f'rank {rank}: injected failure at step {step}')
        recs = torch.full((1, 4, 6 + 4), float(rank * 100 + step))
        cnts = torch.tensor([rank + 1], dtype=torch.int32)
        g_recs, g_cnts = fd.gather_detections(recs, cnts)
        assert g_recs.shape[0] == world and g_cnts[:, 0].tolist() == [r + 1 for r in range(world)]
        assert [float(g_recs[r, 0, 0, 0]) for r in range(world)] == [r * 100.0 + step for r in range(world)]
    dist.destroy_process_group()
    print(f'rank {rank} done', flush=True)


if __name__ == '__main__':
    main()
