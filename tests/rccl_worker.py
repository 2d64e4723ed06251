"""One rank of tests/test_rccl_gpu.py (started by torch.distributed.run, one rank per GPU): the N>1 product path on RCCL —
env shard keyed by the global env id, GraphedIteration (two hipGraph replays with the eager gradient all-reduce between
them) and PipelinedIteration (the all-reduce on the learner's stream beneath the next rollout), identical SharedAdam update on
every replica."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)
    from active_tracking_rl_amd.train import GraphedIteration, PipelinedIteration, default_args, make_player, rollout
    args = default_args(num_envs=256, seed=7)
    player, opt = make_player(args, dev, rank, world)
    assert player.env.core is not None
    # (a) eager: the all-reduced gradient is the mean of the ranks' gradients
    rollout(player, args.num_steps)
    player.compute_grads(opt, args.train_mode)
    local_g = opt.bucket.grad.clone()
    gathered = [torch.zeros_like(local_g) for _ in range(world)]
    dist.all_gather(gathered, local_g)
    player.allreduce_grads(opt)
    mean = sum(gathered) / world
    assert torch.allclose(opt.bucket.grad, mean, rtol=1e-5, atol=1e-7), "all-reduced gradient is not the mean over ranks"
    assert world == 1 or not torch.equal(gathered[0], gathered[1]), "ranks saw the same envs"
    opt.step()
    # (b) graphed: 3 replayed iterations, replicas bit-identical afterwards
    it = GraphedIteration(player, opt, args)
    for _ in range(3):
        it.run()
    torch.cuda.synchronize(dev)
    flats = [torch.zeros_like(opt.bucket.flat) for _ in range(world)]
    dist.all_gather(flats, opt.bucket.flat)
    for f in flats[1:]:
        assert torch.equal(flats[0], f), "replicas diverged"
    assert torch.isfinite(opt.bucket.flat).all()
    st = player.env.core.get_state()
    assert (st["episode"] >= 1).all()
    # (c) the pipelined schedule on the same shard: stream trials + 5 more phases, every rank the same number of updates,
    # replicas bit-identical afterwards (master weights AND both replica weight buffers)
    pit = PipelinedIteration(player, opt, args)
    trials = pit.tune_streams(candidates=1, iters=2)
    assert sum(c for _, c, _ in trials) == 1
    for _ in range(5):
        pit.run()
    pit.finish()
    torch.cuda.synchronize(dev)
    for t in [opt.bucket.flat] + [b.flat for b in pit.buckets]:
        flats = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(flats, t)
        for f in flats[1:]:
            assert torch.equal(flats[0], f), "replicas diverged under the pipelined schedule"
    assert torch.isfinite(opt.bucket.flat).all()
    # (d) the all-reduce captured INSIDE the update graph (ATR_CAPTURE_ALLREDUCE=1; off by default: measured within 1 % of
    # the eager call on one rank, profiles/r04_multirank_1gpu.txt, and never run on a real multi-GPU node): with one rank
    # the collective is forced so that the capture holds it; with more ranks only when asked for (a first run of a captured
    # RCCL collective across GPUs belongs in a session that can watch it)
    if world > 1 and os.environ.get("ATR_TEST_CAPTURED_ALLREDUCE") != "1":
        if rank == 0:
            print("RCCL_OK world=%d elems=%d" % (world, opt.bucket.grad.numel()), flush=True)
        dist.barrier()
        dist.destroy_process_group()
        return
    os.environ["ATR_CAPTURE_ALLREDUCE"] = "1"
    if world == 1:
        os.environ["ATR_FORCE_ALLREDUCE"] = "1"
    try:
        git = GraphedIteration(player, opt, args)
        assert git.capture_allreduce
        for _ in range(3):
            git.run()
        torch.cuda.synchronize(dev)
        flats = [torch.zeros_like(opt.bucket.flat) for _ in range(world)]
        dist.all_gather(flats, opt.bucket.flat)
        for f in flats[1:]:
            assert torch.equal(flats[0], f), "replicas diverged with the captured all-reduce"
        assert torch.isfinite(opt.bucket.flat).all()
    finally:
        os.environ.pop("ATR_CAPTURE_ALLREDUCE", None)
        os.environ.pop("ATR_FORCE_ALLREDUCE", None)
    if rank == 0:
        print("RCCL_OK world=%d elems=%d" % (world, opt.bucket.grad.numel()), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
