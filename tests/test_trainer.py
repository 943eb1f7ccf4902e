import torch

from pytorch_distributed_training_amd.train import TrainConfig, Trainer, run_training


def test_single_process_prints_contract_lines_and_learns(capsys, tmp_path):
    cfg = TrainConfig(use_cpu=True, dataset="synthetic", dataset_length=32, batch_size=8, num_workers=0,
                      learning_rate=1e-3, epochs=6, log_every=1, progress=False,
                      checkpoint_path=str(tmp_path / "m.pt"))
    res = run_training(cfg)
    out = capsys.readouterr().out.splitlines()
    assert out[0] == "Device - cpu"
    assert out[1] == "Training starting..."
    assert "Training complete" in out
    assert any(l.startswith("Elapsed time: ") for l in out)
    assert res.steps == 24 and res.samples_per_sec > 0
    assert sum(res.losses[-4:]) < sum(res.losses[:4])  # memorises 32 images
    assert (tmp_path / "m.pt").exists()


def test_steps_limit_grad_accum_and_resume(tmp_path):
    path = str(tmp_path / "m.pt")
    cfg = TrainConfig(use_cpu=True, dataset="synthetic", dataset_length=64, batch_size=4, num_workers=0,
                      learning_rate=1e-3, steps=3, grad_accum=2, progress=False, checkpoint_path=path)
    tr = Trainer(cfg)
    res = tr.fit()
    assert res.steps == 3 and tr.global_step == 3
    cfg2 = TrainConfig(use_cpu=True, dataset="synthetic", dataset_length=64, batch_size=4, num_workers=0,
                       learning_rate=1e-3, steps=5, progress=False, resume=path)  # --steps counts in total
    tr2 = Trainer(cfg2)
    assert tr2.global_step == 3 and tr2.start_batch == 6  # three steps of two micro-batches into the epoch
    for a, b in zip(tr.raw_model.parameters(), tr2.raw_model.parameters()):
        assert torch.equal(a, b)
    tr2.fit()
    assert tr2.global_step == 5


def test_lm_model_trains_on_cpu():
    cfg = TrainConfig(use_cpu=True, model="gpt2_tiny", seq_len=16, dataset_length=16, batch_size=4, num_workers=0,
                      learning_rate=1e-3, optimizer="adamw", progress=False, log_every=1)
    res = run_training(cfg)
    assert res.steps == 4 and res.final_loss == res.final_loss


def test_incomplete_accumulation_group_at_epoch_end_steps_instead_of_leaking():
    """5 micro-batches per epoch with --grad-accum 2: groups of 2, 2 and 1.  The last, short group is stepped with
    its own loss scale, so every epoch makes 3 steps and starts from clean gradients - compared with plain PyTorch
    doing exactly that."""
    import torch.nn as nn

    from pytorch_distributed_training_amd.data import build_dataset
    from pytorch_distributed_training_amd.models import build_model

    cfg = TrainConfig(use_cpu=True, dataset="synthetic", dataset_length=20, batch_size=4, num_workers=0,
                      learning_rate=0.05, optimizer="sgd", momentum=0.0, weight_decay=0.0, grad_accum=2, epochs=2,
                      progress=False, seed=3)
    tr = Trainer(cfg)
    torch.manual_seed(3)
    ref = build_model("resnet18", 10, 32)
    ref.load_state_dict(tr.raw_model.state_dict())
    res = tr.fit()
    assert res.steps == 6 and tr.global_step == 6

    ds, _ = build_dataset("synthetic", "", "image", 32, 10, 20, seed=3)
    opt = torch.optim.SGD(ref.parameters(), lr=0.05)
    ref.train()
    for _epoch in range(2):
        for group in ([0, 1], [2, 3], [4]):
            opt.zero_grad()
            for b in group:
                xs = torch.stack([ds[i][0] for i in range(b * 4, b * 4 + 4)])
                ys = torch.tensor([ds[i][1] for i in range(b * 4, b * 4 + 4)])
                (nn.functional.cross_entropy(ref(xs), ys) / len(group)).backward()
            opt.step()
    for (n, a), b in zip(tr.raw_model.named_parameters(), ref.parameters()):
        assert torch.allclose(a, b, atol=1e-5), n


def test_resume_continues_behind_the_batches_already_seen(tmp_path):
    """A checkpoint written inside an epoch: the resumed run trains on the remaining batches only and ends with the
    same parameters as an uninterrupted run."""
    path = str(tmp_path / "m.pt")
    base = dict(use_cpu=True, dataset="synthetic", dataset_length=48, batch_size=4, num_workers=0, learning_rate=0.01,
                optimizer="sgd", momentum=0.9, progress=False, seed=5)
    full = Trainer(TrainConfig(**base))
    full.fit()
    assert full.global_step == 12
    first = Trainer(TrainConfig(**base, steps=5, checkpoint_path=path))
    first.fit()
    second = Trainer(TrainConfig(**base, resume="auto", checkpoint_path=path))
    assert second.global_step == 5 and second.start_batch == 5 and second.start_epoch == 0
    res = second.fit()
    assert res.steps == 7 and second.global_step == 12
    for (n, a), b in zip(full.raw_model.named_parameters(), second.raw_model.parameters()):
        assert torch.allclose(a, b, atol=1e-6), n


def test_torn_checkpoint_pair_is_refused(tmp_path):
    import pytest

    from pytorch_distributed_training_amd.train.checkpoint import TornCheckpointError

    path = str(tmp_path / "m.pt")
    base = dict(use_cpu=True, dataset="synthetic", dataset_length=16, batch_size=4, num_workers=0, learning_rate=0.01,
                progress=False)
    Trainer(TrainConfig(**base, steps=2, checkpoint_path=path)).fit()
    side = open(path + ".resume", "rb").read()
    Trainer(TrainConfig(**base, steps=3, checkpoint_path=path)).fit()  # newer weights ...
    open(path + ".resume", "wb").write(side)  # ... next to the older sidecar: a crash between the two writes
    with pytest.raises(TornCheckpointError):
        Trainer(TrainConfig(**base, resume=path))
