"""The seven reference flags keep their names, types and defaults (reference main.py:19-25)."""
import click

from pytorch_distributed_training_amd.cli import main

REFERENCE_FLAGS = {
    "data_dir": ("--data-dir", click.STRING, "~/Downloads/data", False),
    "distributed": ("--distributed", click.BOOL, False, True),
    "use_cpu": ("--use-cpu", click.BOOL, False, True),
    "batch_size": ("--batch-size", click.INT, 32, False),
    "num_workers": ("--num-workers", click.INT, 2, False),
    "learning_rate": ("--learning-rate", click.FLOAT, 0.1, False),
    "weight_decay": ("--weight-decay", click.FLOAT, 0.001, False),
}


def test_reference_flags_present_with_defaults():
    params = {p.name: p for p in main.params}
    for name, (flag, typ, default, is_flag) in REFERENCE_FLAGS.items():
        p = params[name]
        assert flag in p.opts
        assert p.type.name == typ.name
        assert p.default == default
        assert bool(p.is_flag) == is_flag


def test_additive_flags_default_to_reference_workload():
    params = {p.name: p for p in main.params}
    assert params["model"].default == "resnet18"
    assert params["optimizer"].default == "adam"
    assert params["epochs"].default == 1
    assert params["shuffle"].default is False


def test_help_runs():
    from click.testing import CliRunner

    res = CliRunner().invoke(main, ["--help"])
    assert res.exit_code == 0
    assert "--data-dir" in res.output and "[default: ~/Downloads/data]" in res.output
