"""The N>1 path on CPU: two gloo ranks, one flat gradient bucket, ONE all-reduce (SUM) and the
1/world average -- the exchange step of the batch-sharded training.  The HIP
kernels are not involved (they need a GPU); what is checked is that averaging the per-shard
gradients of equal shards reproduces the global-batch gradient and that the bucket plumbing
(views, broadcast of the initial weights, sampler sharding) is right."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from nasa_niswan_amd import ConvLSTM
    from nasa_niswan_amd.optim import FlatParams
    from nasa_niswan_amd.utils import shard_indices
    from oracle import convlstm_oracle as O           # CPU oracle plays the role of the per-rank backward
    torch.manual_seed(rank)                           # deliberately different init per rank ...
    m = ConvLSTM(4, [8], [3], 1)
    flat = FlatParams(m)
    dist.broadcast(flat.data, src=0)                  # ... fixed by the bucket broadcast (FusedTrainer.__init__)
    params = {k: v.clone() for k, v in m.state_dict().items()}
    X, y = O.synth_batch(4, 3, 4, 12, 12, (12, 12), seed=5)      # global batch of 4
    idx = shard_indices(4, 0, rank, world, 2, shuffle=False)[0]
    _, _, _, _, grads = O.train_step(params, None, X[idx], y[idx], lr=1e-3)
    for i, (name, _) in enumerate(m.named_parameters()):
        flat.grad_view(i).copy_(grads[name])
    dist.all_reduce(flat.grad, op=dist.ReduceOp.SUM)  # the ONE collective of the step
    flat.grad.mul_(1.0 / world)                       # (folded into nint_adam_flat's grad_scale on the GPU)
    if rank == 0:
        _, _, _, _, gfull = O.train_step(params, None, X, y, lr=1e-3)
        err = max(float((p.grad - gfull[n]).abs().max() / (gfull[n].abs().max() + 1e-12)) for n, p in m.named_parameters())
        torch.save({"err": err, "w": flat.data.clone()}, out)
    else:
        torch.save({"w": flat.data.clone()}, out + ".1")
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_flat_bucket_allreduce_reproduces_global_gradient(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out), torch.load(out + ".1")
    assert torch.equal(r0["w"], r1["w"])              # identical weights on both ranks after the broadcast
    assert r0["err"] < 1e-5, r0["err"]                # mean of shard grads == global-batch grad (equal shards)
