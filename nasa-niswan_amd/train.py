#!/usr/bin/env python3
"""Fit loop with the reference's command line (reference train.py:148-227) around the fused
MI355X step.  Same flags, same artefacts (`configurations.json`, `logger.npy`,
`epoch-XXX/generator.pth.tar`, per-epoch print), same schedule (Adam betas (0.5,0.999), StepLR
stepped once per epoch before validation, checkpoint every 10 epochs); the per-batch
`loss.item()` / sklearn R2 host syncs (train.py:113-114) are replaced by device accumulators read
once per epoch.  Added flags: --dtype, --levels, --grid, --synthetic-steps, --pad-mode, --f32-inputs.

The data path is on the device too: by default every batch is written by ONE launch of the
fuse / z-score / halo-pad kernel straight into the model's bf16 input slab (dataset.slab_batch);
--f32-inputs materialises the reference's (B,T,C,Hp,Wp) f32 tensor first (dataset.device_batch).

    python nasa-niswan_amd/train.py --model LSTM-demo --in-channels 5 --sequence-length 12 \
        --input-size 100 154 --batch-size 8 --num-epochs 2 --snapshot-dir /tmp/snap
    python -m torch.distributed.run --nproc-per-node 8 nasa-niswan_amd/train.py ...   # batch-sharded DDP
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_arguments(argv=None, MODEL='LSTM-00', SPECIES='bcb', LEARNING_RATE=1.0E-04, DATASET='E33OMA90D', IN_CHANNELS=5,
                  HIDDEN_CHANNELS=(64, 32, 16), KERNEL_SIZE=(5, 3, 3), NUM_LAYERS=3, SEQUENCE_LENGTH=48, TRANSFORM=False,
                  NUM_EPOCHS=50, INPUT_SIZE=(100, 154), BATCH_SIZE=4, NUM_WORKERS=1, SCHEDULER_CONFIG=(10, 0.9),
                  BETAS=(0.5, 0.999), USE_CHECKPOINT=False, SNAPSHOT_DIR='./', RESTORE_FROM='./'):
    """Flag names, types and defaults follow reference train.py:148-208 (INPUT_SIZE defaults to the
    LSTM launcher value instead of the UNet 256x256, and MODEL to an LSTM name: only the LSTM
    family is in scope)."""
    parser = argparse.ArgumentParser(description=f"Training {MODEL} on E33OMA.")
    parser.add_argument("--model", type=str, default=MODEL)
    parser.add_argument("--species", type=str, default=SPECIES)
    parser.add_argument("--learning-rate", type=float, default=LEARNING_RATE)
    parser.add_argument("--dataset", type=str, default=DATASET)
    parser.add_argument("--in-channels", type=int, default=IN_CHANNELS)
    parser.add_argument("--hidden-channels", nargs='+', type=int, default=HIDDEN_CHANNELS)
    parser.add_argument("--kernel-size", nargs='+', type=int, default=KERNEL_SIZE)
    parser.add_argument("--num-layers", type=int, default=NUM_LAYERS)
    parser.add_argument("--sequence-length", type=int, default=SEQUENCE_LENGTH)
    parser.add_argument("--transform", action="store_true", default=TRANSFORM)     # parsed, never read here: the reference passes it to the UNet datasets only (train.py:52-57), its CRNN datasets take no transform (train.py:59-64)
    parser.add_argument("--num-epochs", type=int, default=NUM_EPOCHS)
    parser.add_argument("--input-size", nargs=2, type=int, default=INPUT_SIZE)
    parser.add_argument("--batch-size", type=int, default=BATCH_SIZE)
    parser.add_argument("--num-workers", type=int, default=NUM_WORKERS)     # parsed, unused -- as in the reference (train.py:67)
    parser.add_argument("--scheduler-config", nargs=2, type=float, default=SCHEDULER_CONFIG)
    parser.add_argument("--betas", nargs=2, type=float, default=BETAS)
    parser.add_argument("--use-checkpoint", action="store_true", default=USE_CHECKPOINT)
    parser.add_argument("--snapshot-dir", type=str, default=SNAPSHOT_DIR)
    parser.add_argument("--restore-from", type=str, default=RESTORE_FROM)
    # extensions
    parser.add_argument("--dtype", type=str, default="bf16", choices=["bf16", "f32"])
    parser.add_argument("--levels", type=int, default=1, help="vertical levels fused as channels (C = 3L+2)")
    parser.add_argument("--synthetic-steps", type=int, default=480, help="length of the synthetic record")
    parser.add_argument("--pad-mode", type=str, default="reference", choices=["reference", "reflect"])
    parser.add_argument("--grid", nargs=2, type=int, default=(90, 144),
                        help="un-padded lat x lon grid of the synthetic record; the loss crop is (input-size - grid)/2 "
                             "(the reference hard-codes 90x144 and halo 5, train.py:102)")
    parser.add_argument("--f32-inputs", action="store_true",
                        help="materialise X as the reference's f32 (B,T,C,Hp,Wp) tensor instead of writing the input slab directly")
    args = parser.parse_args(argv)
    rank = int(os.environ.get("RANK", "0"))
    if rank == 0:
        os.makedirs(args.snapshot_dir, exist_ok=True)
        print('Working Directory:', args.snapshot_dir)
        with open(os.path.join(args.snapshot_dir, 'configurations.json'), "w") as f:     # train.py:221-225
            json.dump(vars(args), f, indent=4)
    return args


def main(args):
    since = time.time()
    import torch.distributed as dist
    import torch.optim as optim
    import nasa_niswan_amd as pkg
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    from nasa_niswan_amd.trainer import FusedTrainer
    from nasa_niswan_amd.utils import load_checkpoint, save_checkpoint, seed, shard_indices

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("train.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    if rank == 0:
        print(f"{args.model} is deployed on {torch.cuda.get_device_name(local_rank)}")     # train.py:29
    seed(0)                                                                                 # train.py:32
    if args.model.split('-')[0] != 'LSTM':
        raise SystemExit("only the LSTM family (ConvLSTM) is in scope of this build")
    out_ch = args.levels
    generator = pkg.ConvLSTM(args.in_channels, list(args.hidden_channels), list(args.kernel_size), args.num_layers,
                             out_channels=out_ch, compute_dtype=args.dtype).to(dev)         # train.py:48
    H, W = (int(v) for v in args.grid)
    if args.input_size[0] < H or args.input_size[1] < W:
        raise SystemExit(f"--input-size {tuple(args.input_size)} is smaller than --grid {(H, W)}: the model runs on the "
                         "grid plus its halo (launcher.sh:24)")
    halo = ((args.input_size[0] - H) // 2, (args.input_size[1] - W) // 2)                    # 5,5 in the reference (train.py:102)
    ds_kw = dict(species=args.species, padding=tuple(args.input_size), in_channels=args.in_channels,
                 sequence_length=args.sequence_length, levels=args.levels, n_steps=args.synthetic_steps,
                 grid=(H, W), pad_mode=args.pad_mode, device=dev)
    train_dataset = SyntheticE33OMA_CRNN('train', **ds_kw)                                   # train.py:63-65
    val_dataset = SyntheticE33OMA_CRNN('val', **ds_kw)
    get_batch = (lambda ds, idx: ds.device_batch(idx)) if args.f32_inputs else (lambda ds, idx: ds.slab_batch(idx))

    trainer = FusedTrainer(generator, lr=args.learning_rate, betas=tuple(args.betas), halo=halo)        # train.py:71
    optimizer = trainer.optimizer
    scheduler = optim.lr_scheduler.StepLR(optimizer, step_size=int(args.scheduler_config[0]),
                                          gamma=args.scheduler_config[1])                   # train.py:72
    if args.use_checkpoint:
        load_checkpoint(f'{args.restore_from}/generator.pth.tar', generator, optimizer, args.learning_rate,
                        map_location=dev)                                                   # train.py:77-78
    logger = {'MSELoss': [], 'r2_score': [], 'r2_score_val': [], 'first_step_loss': None, 'train_samples_per_s': []}
    for epoch in range(1, args.num_epochs + 1):                                              # train.py:82
        generator.train()
        trainer.reset_stats()
        torch.cuda.synchronize()
        t_epoch, n_epoch = time.time(), 0
        for idx in shard_indices(len(train_dataset), epoch, rank, world, args.batch_size):  # train.py:89
            X, y = get_batch(train_dataset, idx)                                             # preproc on device
            loss = trainer.step(X, y)                                                        # train.py:96-110
            n_epoch += len(idx)
            if logger['first_step_loss'] is None:
                logger['first_step_loss'] = float(loss)
        loss_e, r2_e = trainer.epoch_stats()                                                 # one host read per epoch (syncs)
        logger['train_samples_per_s'].append(world * n_epoch / max(time.time() - t_epoch, 1e-9))   # data path included
        logger['MSELoss'].append(loss_e)                                                     # (MSE+L1, as in train.py:116)
        logger['r2_score'].append(r2_e)
        scheduler.step()                                                                     # train.py:120
        generator.eval()
        trainer.reset_stats()
        for idx in shard_indices(len(val_dataset), 0, rank, world, 1, shuffle=False):        # utils.py:52-75, batch 1
            trainer.evaluate(*get_batch(val_dataset, idx))
        logger['r2_score_val'].append(trainer.epoch_stats()[1])
        if rank == 0:
            print(f"Epoch: {epoch}, Loss: {logger['MSELoss'][-1]:.5f}, R2T: {logger['r2_score'][-1]:.5f}, "
                  f"R2V: {logger['r2_score_val'][-1]:.5f}")                                  # train.py:124
            print(f"  train loop incl. device preproc: {logger['train_samples_per_s'][-1]:.1f} samples/s")
            if epoch % 10 == 0:                                                              # train.py:126-136
                d = os.path.join(args.snapshot_dir, f'epoch-{epoch:003d}')
                os.makedirs(d, exist_ok=True)
                print('Learning Rate:', scheduler.get_last_lr())
                save_checkpoint(generator, optimizer, os.path.join(d, 'generator.pth.tar'), scheduler.get_last_lr(), epoch)
    if rank == 0:
        with open(os.path.join(args.snapshot_dir, "logger.npy"), mode='wb') as f:           # train.py:138-142
            np.save(f, np.array(logger['MSELoss']))
            np.save(f, np.array(logger['r2_score']))
            np.save(f, np.array(logger['r2_score_val']))
        time_elapsed = time.time() - since
        print(f'Training complete in {time_elapsed // 60:.0f}m {time_elapsed % 60:.0f}s')
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return logger


if __name__ == '__main__':
    main(get_arguments())
