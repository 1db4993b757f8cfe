"""Times torch.distributed's gloo collectives on DEVICE tensors between two processes sharing cuda:0 -- the transport the
multi-process GPU tests and bench.py's gloo:device mode use.  Explains those runs' step times (they are not measurements of
the layouts): python tools/microbench/gloo_device_collectives.py"""
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def work(rank, world, port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    for name, n in (("1.3 MB", 327680), ("17.8 MB", 4461824)):
        inp = torch.full((n,), float(rank), device="cuda")
        out = torch.empty(world * n, device="cuda")
        for label, fn in (("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(out, inp)),
                          ("all_reduce", lambda: dist.all_reduce(inp)),
                          ("all_to_all_single", lambda: dist.all_to_all_single(out[:n], inp))):
            try:
                fn(); torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5
                if rank == 0:
                    print(f"{label:26s} {name:8s} {1e3 * dt:9.3f} ms", flush=True)
            except Exception as e:                           # noqa: BLE001
                if rank == 0:
                    print(f"{label:26s} {name:8s} failed: {e!r}"[:200], flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    mp.spawn(work, args=(2, int(sys.argv[1]) if len(sys.argv) > 1 else 29533), nprocs=2, join=True)
