"""SURVEY.md 8(e) on the device with REAL processes: world-2 / world-4 jobs whose ranks all sit on the one GPU of the box and
talk over gloo with device tensors (RCCL refuses two ranks on one device; the collectives' call sites, group construction,
stream ordering against the HIP kernels and the two-graphs-around-a-collective capture are the same code either way).
Same workers and the same oracle checks as the CPU gloo tests (tests/test_dist_cpu.py), with the HIP library instead of
the CPU stand-ins; plus captured-graph steps against eager steps for the layouts bench.py runs at N > 1."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.test_dist_cpu import (_dp_worker, _free_port, _gloo_moves_device_tensors, _worker, check_against_oracle,
                                 check_dp_against_oracle)

pytestmark = pytest.mark.gpu


def _skip_without_device_gloo(tmp_path):
    flag = os.path.join(str(tmp_path), "no_device_gloo")
    if os.path.exists(flag):
        pytest.skip("this torch build's gloo does not move device tensors: " + open(flag).read()[:200])


@pytest.mark.parametrize("model,world,layout", [("XSimGCL", 2, "cols"), ("XSimGCL", 2, "rows"), ("XSimGCL", 4, "2d"),
                                                ("SGL", 2, "cols"), ("LightGCN", 3, "rows")])
def test_sharded_processes_on_the_device_equal_the_oracle(tmp_path, model, world, layout):
    out = str(tmp_path / "res.npz")
    mp.spawn(_worker, args=(world, _free_port(), model, layout, out, 64, "cuda"), nprocs=world, join=True)
    _skip_without_device_gloo(tmp_path)
    check_against_oracle(out, model, 64, atol=2e-5)


@pytest.mark.parametrize("model,world", [("XSimGCL", 2), ("MF", 3)])
def test_data_parallel_processes_on_the_device_equal_the_mean_gradient_oracle(tmp_path, model, world):
    mp.spawn(_dp_worker, args=(world, _free_port(), model, str(tmp_path), "cuda"), nprocs=world, join=True)
    _skip_without_device_gloo(tmp_path)
    check_dp_against_oracle(tmp_path, model, world)


def _graph_worker(rank, world, port, layout, out_dir):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if not _gloo_moves_device_tensors(out_dir, rank):
        dist.destroy_process_group()
        return
    from selfrec_amd import synth
    from selfrec_amd.data.ui_graph import Interaction
    from selfrec_amd.dist import ShardedTrainer
    tu, ti, su, si, U, I = synth.make_dataset("tiny")
    data = Interaction({}, synth.as_triples(tu, ti), [])
    res = {}
    for graphed in (False, True):
        torch.manual_seed(0)                       # same initial tables; in-kernel counter RNG for the perturbation
        tr = ShardedTrainer(data, 64, model="XSimGCL", n_layers=3, batch_size=500, layer_cl=1, tau=0.2, eps=0.2, cl_rate=0.2,
                            use_graph=graphed, device="cuda", layout=layout)
        tr.seed_sampler(40)
        tr.begin_epoch()
        per_step = []
        for _ in range(5):                         # (the first graphed step captures, the others replay)
            tr.step()
            per_step.append(tr.E0.detach().clone())
        torch.cuda.synchronize()
        if layout in ("dp", "cols"):               # the collective sits BETWEEN the two graphs: the capture must succeed
            assert tr.use_graph == graphed and (tr._graph is not None) == graphed
        # (2-D / rows keep collectives inside the step: a capture that the backend refuses falls back to eager launches --
        #  engine.step_phases -- and the run must still be right)
        pu, pi = tr.parameters_full()
        res[graphed] = (torch.cat([pu, pi]).cpu().numpy(), np.asarray(tr.read_losses()), [t.cpu().numpy() for t in per_step])
    # captured steps == eager steps up to the loss section's float atomics (their order differs from run to run:
    # profiles/r03_c_determinism.txt), amplified by five Adam steps
    diff = np.abs(res[False][0] - res[True][0])
    by_step = [float(np.abs(a - b).max()) for a, b in zip(res[False][2], res[True][2])]
    assert diff.max() < 2e-5 and np.median(diff) < 1e-7, (rank, float(diff.max()), float(np.median(diff)), by_step)
    np.testing.assert_allclose(res[False][1], res[True][1], rtol=2e-5)
    np.save(os.path.join(out_dir, f"params{rank}.npy"), res[True][0])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.selfcheck
@pytest.mark.parametrize("layout,world", [("dp", 2), ("cols", 2), ("2d", 4)])
def test_captured_steps_around_the_collective_equal_eager_steps(tmp_path, layout, world):
    """bench.py at N > 1: two hipGraphs per step with the collective between them (engine._capture), here with more than one
    real rank.  The replayed steps leave the parameters of the eager ones, and every rank ends with the same full tables."""
    mp.spawn(_graph_worker, args=(world, _free_port(), layout, str(tmp_path)), nprocs=world, join=True)
    _skip_without_device_gloo(tmp_path)
    tables = [np.load(tmp_path / f"params{r}.npy") for r in range(world)]
    for t in tables[1:]:
        assert np.array_equal(t, tables[0])             # (replicas / gathered tables: the same bits on every rank)
