"""CPU tests of the host-side mirror of the reference interface: CLI flags, config dump,
sampler sharding, synthetic dataset windows/splits, flat bucket, optimizer/checkpoint formats."""
import json
import os

import numpy as np
import pytest
import torch


def test_get_arguments_mirrors_reference_flags(tmp_path, monkeypatch):
    from nasa_niswan_amd.train import get_arguments
    monkeypatch.delenv("RANK", raising=False)
    snap = tmp_path / "snap"
    # the canonical invocation of the reference (launcher.sh:13-30), trimmed to in-scope values
    args = get_arguments(["--model", "LSTM-64K5.32K3.16K3-E33OMA-5C-BCB", "--species", "bcb", "--learning-rate", "1.0E-03",
                          "--dataset", "E33OMA", "--in-channels", "5", "--hidden-channels", "64", "32", "16",
                          "--kernel-size", "5", "3", "3", "--num-layers", "3", "--sequence-length", "48",
                          "--num-epochs", "30", "--input-size", "100", "154", "--batch-size", "8", "--num-workers", "1",
                          "--scheduler-config", "10", "0.9", "--betas", "0.5", "0.999", "--snapshot-dir", str(snap),
                          "--restore-from", str(snap)])
    assert args.hidden_channels == [64, 32, 16] and args.kernel_size == [5, 3, 3] and args.betas == [0.5, 0.999]
    assert args.input_size == [100, 154] and args.scheduler_config == [10.0, 0.9]
    cfg = json.load(open(snap / "configurations.json"))          # train.py:221-225
    for key in ["model", "species", "learning_rate", "dataset", "in_channels", "hidden_channels", "kernel_size",
                "num_layers", "sequence_length", "transform", "num_epochs", "input_size", "batch_size", "num_workers",
                "scheduler_config", "betas", "use_checkpoint", "snapshot_dir", "restore_from"]:
        assert key in cfg
    # reference defaults (train.py:148-168) that survive
    d = get_arguments(["--snapshot-dir", str(snap)])
    assert d.learning_rate == 1e-4 and tuple(d.betas) == (0.5, 0.999) and d.batch_size == 4 and d.num_epochs == 50
    assert tuple(d.hidden_channels) == (64, 32, 16) and tuple(d.kernel_size) == (5, 3, 3) and d.sequence_length == 48


def test_shard_indices_partitions_each_global_batch():
    from nasa_niswan_amd.utils import shard_indices
    n, bs, world = 103, 4, 2
    per_rank = [shard_indices(n, 3, r, world, bs) for r in range(world)]
    assert len(per_rank[0]) == len(per_rank[1]) == n // (bs * world)
    seen = np.concatenate([np.concatenate(p) for p in per_rank])
    assert len(seen) == len(set(seen.tolist())) == (n // 8) * 8          # disjoint, ragged tail dropped
    # world=1 sees the same global batches in the same order
    one = shard_indices(n, 3, 0, 1, bs * world)
    for s in range(len(per_rank[0])):
        np.testing.assert_array_equal(one[s], np.concatenate([per_rank[0][s], per_rank[1][s]]))
    # a single rank keeps the ragged tail batch, as the reference's DataLoader(drop_last=False) does (train.py:67)
    assert len(one) == n // 8 + 1 and len(one[-1]) == n % 8
    assert sorted(np.concatenate(one).tolist()) == list(range(n))
    assert not np.array_equal(np.concatenate(shard_indices(n, 4, 0, 1, 8)), np.concatenate(one))   # reshuffled per epoch
    np.testing.assert_array_equal(np.concatenate(shard_indices(10, 0, 0, 1, 1, shuffle=False)), np.arange(10))


def test_synthetic_dataset_windows_and_splits():
    from nasa_niswan_amd.dataset import SyntheticE33OMA_CRNN
    kw = dict(padding=(100, 154), in_channels=5, sequence_length=12, n_steps=100, device="cpu")
    tr, va, te = (SyntheticE33OMA_CRNN(p, **kw) for p in ("train", "val", "test"))
    assert (len(tr), len(va)) == (70, 10) and len(te) == 100 - 12 + 1 - 80      # 70/10/rest (dataset.py:601-612)
    (u, v, w, pr, src), y = tr.window(3)
    assert u.shape == (12, 1, 90, 144) and pr.shape == (12, 90, 144) and y.shape == (1, 90, 144)
    # the target is the tracer at the window's LAST step (dataset.py:599)
    np.testing.assert_array_equal(y, tr.yraw[3 + 12 - 1])
    np.testing.assert_array_equal(u, tr.u[3:15])
    assert pr.min() >= 0 and src.min() >= 0
    # statistics come from the training part of the record only (dataset.py:589-596)
    np.testing.assert_allclose(tr.X_mean[0], tr.u[:70, 0].mean(), rtol=1e-6)
    assert tr.X_mean.shape == (5,) and va.first[0] == 70
    lv = SyntheticE33OMA_CRNN("train", padding=(100, 154), in_channels=14, sequence_length=4, levels=4, n_steps=40, device="cpu")
    assert lv.X_mean.shape == (14,)
    # a channel count that is not 3L+2 (BASELINE configs[0]: 4 channels on 32x32) synthesises generic fields;
    # the reference's static-attribute channels (dataset.py:100-122) stay out of scope
    gen = SyntheticE33OMA_CRNN("train", padding=(32, 32), in_channels=4, sequence_length=4, n_steps=24, grid=(32, 32), device="cpu")
    assert gen.generic and gen.X_mean.shape == (4,) and len(gen) == 17
    (f,), y = gen.window(2)
    assert f.shape == (4, 4, 32, 32) and y.shape == (1, 32, 32)


def _tiny():
    from nasa_niswan_amd import ConvLSTM
    torch.manual_seed(0)
    return ConvLSTM(4, [8], [3], 1)


def test_flat_bucket_aliases_parameters_and_grads():
    from nasa_niswan_amd.optim import FlatParams
    m = _tiny()
    before = {k: v.clone() for k, v in m.state_dict().items()}
    flat = FlatParams(m)
    assert flat.numel == sum(p.numel() for p in m.parameters()) == 3497        # BASELINE.md cfg 0
    for k, v in m.state_dict().items():
        assert torch.equal(v, before[k])
    flat.data.mul_(2.0)                                   # the bucket IS the parameters
    assert torch.equal(m.state_dict()["conv.weight"], 2 * before["conv.weight"])
    flat.grad.fill_(3.0)
    assert all(float(p.grad.min()) == 3.0 for p in m.parameters())
    assert flat.is_intact()
    m.load_state_dict(before)                             # in-place copy keeps the aliasing
    assert flat.is_intact() and torch.equal(flat.data[:before["layers.0.conv.weight"].numel()],
                                            before["layers.0.conv.weight"].reshape(-1))


def test_fused_adam_state_dict_is_torch_adam_format(tmp_path):
    from nasa_niswan_amd.optim import FlatParams, FusedAdam
    from nasa_niswan_amd.utils import load_checkpoint, save_checkpoint
    m = _tiny()
    opt = FusedAdam(FlatParams(m), lr=1e-3, betas=(0.5, 0.999))
    ref = torch.optim.Adam(_tiny().parameters(), lr=1e-3, betas=(0.5, 0.999))
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == ref.state_dict()["param_groups"][0]["params"] == [0, 1, 2, 3]
    for key in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
        assert sd["param_groups"][0][key] == ref.state_dict()["param_groups"][0][key]
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    # a checkpoint written by torch's Adam (the reference's format, utils.py:23-32) loads into FusedAdam
    m2 = _tiny()
    for p in m2.parameters():
        p.grad = torch.ones_like(p)
    ref2 = torch.optim.Adam(m2.parameters(), lr=1e-3, betas=(0.5, 0.999))
    ref2.step(); ref2.step()
    path = tmp_path / "generator.pth.tar"
    save_checkpoint(m2, ref2, str(path), [5e-4], 20)
    ck = load_checkpoint(str(path), m, opt, lr=2e-3)
    assert ck["epoch"] == 20 and opt.param_groups[0]["lr"] == 2e-3           # LR forced to the CLI value (utils.py:44-46)
    assert opt._step == 2
    off = opt.flat.offsets[1]
    np.testing.assert_allclose(opt.exp_avg[off:off + 4].numpy(), ref2.state[list(m2.parameters())[1]]["exp_avg"][:4].numpy())
    assert torch.equal(m.state_dict()["conv.weight"], m2.state_dict()["conv.weight"])
    # StepLR drives it like any optimizer (train.py:72,120)
    sch = torch.optim.lr_scheduler.StepLR(opt, step_size=2, gamma=0.5)
    for _ in range(4):
        sch.step()
    assert abs(opt.param_groups[0]["lr"] - 2e-3 * 0.25) < 1e-12
    # and a FusedAdam checkpoint loads back into torch's Adam
    ref3 = torch.optim.Adam(_tiny().parameters(), lr=1e-3)
    ref3.load_state_dict(opt.state_dict())
    assert float(ref3.state_dict()["state"][0]["step"]) == 2.0


def test_module_tree_matches_reference_state_dict():
    """parameter names, shapes and count of the reference stack
    (model.py:197-251; 580,305 parameters, test.ipynb:4698-4699); AssertionError of model.py:237."""
    import pytest
    import nasa_niswan_amd as pkg
    net = pkg.ConvLSTM(5, [64, 32, 16], [5, 3, 3], 3)
    sd = net.state_dict()
    want = {"layers.0.conv.weight": (256, 69, 5, 5), "layers.0.conv.bias": (256,),
            "layers.1.conv.weight": (128, 96, 3, 3), "layers.1.conv.bias": (128,),
            "layers.2.conv.weight": (64, 48, 3, 3), "layers.2.conv.bias": (64,),
            "conv.weight": (1, 16, 1, 1), "conv.bias": (1,)}
    assert {k: tuple(v.shape) for k, v in sd.items()} == want
    assert sum(p.numel() for p in net.parameters()) == 580305
    cell = net.layers[0]
    for attr in ("input_channels", "hidden_channels", "kernel_size", "padding", "bias", "conv", "sigmoid", "tanh"):
        assert hasattr(cell, attr), attr
    with pytest.raises(AssertionError):
        pkg.ConvLSTM(5, [64, 32], [5, 3, 3], 3)


def test_from_arrays_reproduces_the_reference_in_memory_dataset():
    """f-3 as the survey words it: `E33OMA90D_CRNN` (dataset.py:551-637) on arrays the caller holds.  A 4320-step record
    (the reference's 90 days x 48) on a tiny grid: split 3023 / 432 / rest, statistics over the first 3023 steps, windows and
    target lag against the numpy restatement of dataset.py:584-616."""
    import numpy as np
    from nasa_niswan_amd.dataset import E33OMA90D_CRNN, reference_split
    from oracle import preproc_oracle as PO
    assert reference_split(4320) == (3023, 3455)
    rng = np.random.default_rng(3)
    n, H, W, T = 4320, 4, 6, 7
    arrs = [(rng.standard_normal((n, H, W)) * s + m).astype(np.float32) for m, s in ((0.2, 6.5), (0.3, 5.3), (0, 6e-5), (2.2, 7.3), (0.2, 2.6), (5.0, 57.0))]
    sizes = {}
    for period in ("train", "val", "test"):
        ds = E33OMA90D_CRNN.from_arrays(*arrs, period=period, padding=None, sequence_length=T, device="cpu")
        Xr, yr, Xm, Xs, ym, ys = PO.inmemory_rnn_dataset(*arrs, period, T)
        sizes[period] = len(ds)
        assert len(ds) == len(yr) == len(Xr)
        np.testing.assert_allclose(ds.X_mean, Xm.reshape(-1), rtol=1e-6)
        np.testing.assert_allclose(ds.X_std, Xs.reshape(-1), rtol=1e-6)
        np.testing.assert_allclose(ds.y_mean, ym.reshape(()), rtol=1e-6)
        np.testing.assert_allclose(ds.y_std, ys.reshape(()), rtol=1e-6)
        for i in (0, len(ds) // 2, len(ds) - 1):
            fields, ylast = ds.window(i)
            raw = np.stack([f[:, 0] if f.ndim == 4 else f for f in fields], axis=1)             # (T, 5, H, W)
            np.testing.assert_allclose((raw - Xm) / Xs, Xr[i], rtol=1e-5, atol=1e-6)
            np.testing.assert_allclose((ylast[0] - ym[0]) / ys[0], yr[i], rtol=1e-5, atol=1e-6)
    assert sizes == {"train": 3023, "val": 432, "test": 4320 - T + 1 - 3455}
    with pytest.raises(TypeError, match="from_arrays"):
        E33OMA90D_CRNN("train", "bcb", (100, 154))
    with pytest.raises(ValueError, match="prec"):
        E33OMA90D_CRNN.from_arrays(arrs[0], arrs[1], arrs[2], arrs[3][:, :2], arrs[4], arrs[5], device="cpu")
