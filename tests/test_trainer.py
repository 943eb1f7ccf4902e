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
                       learning_rate=1e-3, steps=2, progress=False, resume=path)
    tr2 = Trainer(cfg2)
    assert tr2.global_step == 3
    for a, b in zip(tr.raw_model.parameters(), tr2.raw_model.parameters()):
        assert torch.equal(a, b)
    tr2.fit()
    assert tr2.global_step == 5


def test_lm_model_trains_on_cpu():
    cfg = TrainConfig(use_cpu=True, model="gpt2_tiny", seq_len=16, dataset_length=16, batch_size=4, num_workers=0,
                      learning_rate=1e-3, optimizer="adamw", progress=False, log_every=1)
    res = run_training(cfg)
    assert res.steps == 4 and res.final_loss == res.final_loss
