"""Do captured data-parallel steps (two hipGraphs around the all-reduce) equal eager ones when there are 2 real ranks?
tests/test_gpu_multiproc.py saw one divergent run in two.  Repeats the comparison with host synchronisations inserted
before / after the collective to find which ordering, if any, is not guaranteed:

    python tools/dp_graph_probe.py [trials]
"""
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def work(rank, world, port, trials):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from selfrec_amd import synth
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.dist import ShardedTrainer
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])

    def run(graphed, sync_before, sync_after, steps=5):
        torch.manual_seed(0)
        tr = ShardedTrainer(data, 64, model="XSimGCL", n_layers=3, batch_size=500, layer_cl=1, tau=0.2, eps=0.2, cl_rate=0.2,
                            use_graph=graphed, device="cuda", layout="dp")
        tr.seed_sampler(40)
        tr.begin_epoch()
        real = tr._dp_allreduce

        def wrapped():
            if sync_before:
                torch.cuda.synchronize()
            real()
            if sync_after:
                torch.cuda.synchronize()
        tr._dp_allreduce = wrapped
        traj = []
        for _ in range(steps):
            tr.step()
            traj.append(tr.E0.detach().clone())
        torch.cuda.synchronize()
        out = [t.cpu().numpy() for t in traj]
        tr._dp_allreduce = None                    # (break the cycle: a CUDAGraph must not be destroyed by the cyclic GC
        del tr, real, wrapped                      #  in the middle of the next trainer's capture)
        import gc
        gc.collect()
        return out

    ref = run(False, True, True)
    again = run(False, True, True)
    noise = max(float(np.abs(a - b).max()) for a, b in zip(ref, again))
    if rank == 0:
        print(f"eager vs eager (both fully synchronised): max |diff| over 5 steps {noise:.2e}", flush=True)
    for name, graphed, sb, sa in (("eager, no host sync", False, False, False), ("graphs, no host sync", True, False, False),
                                  ("graphs, sync before", True, True, False), ("graphs, sync after", True, False, True),
                                  ("graphs, sync both", True, True, True)):
        bad, first, worst = 0, [], 0.0
        for t in range(trials):
            got = run(graphed, sb, sa)
            d = [float(np.abs(a - b).max()) for a, b in zip(ref, got)]
            worst = max(worst, max(d))
            if max(d) > 1e-4:
                bad += 1
                first.append(next(k for k, v in enumerate(d) if v > 1e-4))
        flag = torch.tensor([float(bad)], device="cuda")
        dist.all_reduce(flag)
        if rank == 0:
            print(f"{name:22s}: {bad}/{trials} divergent trials on rank 0 ({int(flag.item())} over both ranks), first divergent "
                  f"step {first}, worst |diff| {worst:.2e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    t0 = time.time()
    mp.spawn(work, args=(2, 29541, trials), nprocs=2, join=True)
    print(f"{time.time() - t0:.1f} s")
