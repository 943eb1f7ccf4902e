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


def test_torn_checkpoint_pair_is_refused_when_no_generation_is_kept(tmp_path):
    import pytest

    from pytorch_distributed_training_amd.train.checkpoint import TornCheckpointError

    path = str(tmp_path / "m.pt")
    base = dict(use_cpu=True, dataset="synthetic", dataset_length=16, batch_size=4, num_workers=0, learning_rate=0.01,
                progress=False, keep_checkpoints=0)
    Trainer(TrainConfig(**base, steps=2, checkpoint_path=path)).fit()
    side = open(path + ".resume", "rb").read()
    Trainer(TrainConfig(**base, steps=3, checkpoint_path=path)).fit()  # newer weights ...
    open(path + ".resume", "wb").write(side)  # ... next to the older sidecar: a crash between the two writes
    with pytest.raises(TornCheckpointError):
        Trainer(TrainConfig(**base, resume=path))
    assert Trainer(TrainConfig(**base, resume="auto", checkpoint_path=path)).global_step == 0  # nothing whole: fresh start


def test_torn_newest_checkpoint_falls_back_to_the_generation_before(tmp_path, capsys):
    import os

    from pytorch_distributed_training_amd.train import checkpoint as ckpt

    path = str(tmp_path / "m.pt")
    base = dict(use_cpu=True, dataset="synthetic", dataset_length=32, batch_size=4, num_workers=0, learning_rate=0.01,
                progress=False)
    Trainer(TrainConfig(**base, steps=2, checkpoint_path=path)).fit()
    Trainer(TrainConfig(**base, steps=3, checkpoint_path=path, resume="auto")).fit()
    Trainer(TrainConfig(**base, steps=5, checkpoint_path=path, resume="auto")).fit()
    gens = ckpt.list_generations(path)
    assert [g[0] for g in gens] == [5, 3]  # --keep-checkpoints 2: generation 2 was pruned
    assert ckpt.pair_state(path) == "intact" and ckpt.find_resumable(path) == path
    # the job dies after the newest generation's model file and before its sidecar, having already ... nothing else
    os.remove(ckpt.resume_path(gens[0][1]))
    # ... and, in another scenario, between re-pointing the main model file and the main sidecar
    os.replace(gens[1][1] + ".resume", path + ".resume")
    ckpt._link_over(path + ".resume", gens[1][1] + ".resume")
    assert ckpt.pair_state(path) == "torn" and ckpt.pair_state(gens[0][1]) == "weights_only"
    assert ckpt.find_resumable(path) == gens[1][1]
    capsys.readouterr()
    tr = Trainer(TrainConfig(**base, steps=6, checkpoint_path=path, resume="auto"))
    out = capsys.readouterr().out
    assert tr.global_step == 3 and "using the intact generation" in out and "step000000003" in out
    tr.fit()
    assert tr.global_step == 6 and ckpt.pair_state(path) == "intact"


def test_checkpoint_on_the_last_batch_of_an_epoch_resumes_with_the_next_epoch(tmp_path):
    """--steps equal to a whole number of epochs: the checkpoint is the start of the next epoch, so the resumed run
    neither replays the finished epoch nor trains one too many, and ends where an uninterrupted run ends."""
    path = str(tmp_path / "m.pt")
    base = dict(use_cpu=True, dataset="synthetic", dataset_length=16, batch_size=4, num_workers=0, learning_rate=0.01,
                optimizer="sgd", momentum=0.9, progress=False, seed=7, shuffle=True, epochs=3)
    full = Trainer(TrainConfig(**base))
    full.fit()
    assert full.global_step == 12
    first = Trainer(TrainConfig(**base, steps=8, checkpoint_path=path))  # stops on the last batch of epoch 1
    first.fit()
    second = Trainer(TrainConfig(**base, resume="auto", checkpoint_path=path))
    assert (second.start_epoch, second.start_batch, second.global_step) == (2, 0, 8)
    res = second.fit()
    assert res.steps == 4 and res.epochs == 1 and second.global_step == 12
    for (n, a), b in zip(full.raw_model.named_parameters(), second.raw_model.parameters()):
        assert torch.allclose(a, b, atol=1e-6), n
    # the same through --save-every landing on the boundary
    path2 = str(tmp_path / "n.pt")
    Trainer(TrainConfig(**base, steps=9, save_every=4, checkpoint_path=path2)).fit()
    from pytorch_distributed_training_amd.train import checkpoint as ckpt

    side = torch.load(ckpt.resume_path(ckpt.generation_path(path2, 8)), weights_only=True)
    assert (side["epoch"], side["extra"]["batch_in_epoch"]) == (2, 0)


def test_evaluation_counts_every_sample_once_for_any_world_size():
    """10 samples on 3 ranks: the sampler pads the shards to 4 samples each by wrapping around; the totals must still be
    those of the 10 samples."""
    from pytorch_distributed_training_amd.parallel import DistEnv

    base = dict(use_cpu=True, dataset="synthetic", dataset_length=10, batch_size=3, num_workers=0, progress=False, seed=2)
    one = Trainer(TrainConfig(**base))
    ref = one.evaluate_totals()
    assert ref[2] == 10
    parts = []
    for r in range(3):
        tr = Trainer(TrainConfig(**base), env=DistEnv(rank=r, world_size=3))
        tr.raw_model.load_state_dict(one.raw_model.state_dict())
        parts.append(tr.evaluate_totals())
    assert [p[2] for p in parts] == [4, 3, 3]
    tot = [sum(p[i] for p in parts) for i in range(3)]
    assert tot[2] == 10 and tot[1] == ref[1] and abs(tot[0] - ref[0]) < 1e-4 * max(1.0, abs(ref[0]))
