"""Flag registry and path helpers (surface of /root/reference/example.py:56-105)."""
import pytest

from distributed_tensorflow_amd import flags as fl
from distributed_tensorflow_amd import paths


def fresh():
    return fl.FlagValues()


def test_defaults_are_type_converted_like_tf_flags():
    fv = fresh()
    fl.DEFINE_integer("task_index", "0", "env string default", flag_values=fv)   # TASK_INDEX arrives as a string
    fl.DEFINE_string("job_name", None, "", flag_values=fv)
    fv([])
    assert fv.task_index == 0 and isinstance(fv.task_index, int)
    assert fv.job_name is None
    assert (fv.task_index == 0) is True          # what `is_chief=(FLAGS.task_index == 0)` relies on


def test_command_line_overrides_defaults_and_ignores_unknown():
    fv = fresh()
    fl.DEFINE_string("job_name", "worker", "", flag_values=fv)
    fl.DEFINE_integer("task_index", 0, "", flag_values=fv)
    fl.DEFINE_string("ps_hosts", None, "", flag_values=fv)
    fl.DEFINE_float("lr", 0.1, "", flag_values=fv)
    fl.DEFINE_boolean("sync", False, "", flag_values=fv)
    rest = fv(["--job_name=ps", "--task_index", "3", "--ps_hosts=a:1,b:2", "--sync", "--unknown=1", "pos", "--lr=0.5"])
    assert (fv.job_name, fv.task_index, fv.ps_hosts, fv.lr, fv.sync) == ("ps", 3, "a:1,b:2", 0.5, True)
    assert rest == ["--unknown=1", "pos"]
    fv(["--nosync", "--sync=false"])
    assert fv.sync is False
    with pytest.raises(AttributeError):
        _ = fv.nope
    with pytest.raises(ValueError):
        fl.DEFINE_string("job_name", "x", "", flag_values=fv)
    with pytest.raises(ValueError):
        fv(["--task_index", "notanint"])


def test_lazy_parse_reads_sys_argv(monkeypatch):
    fv = fresh()
    fl.DEFINE_string("log_dir", "/tmp/x", "", flag_values=fv)
    monkeypatch.setattr("sys.argv", ["prog", "--log_dir", "/data/logs"])
    assert fv.log_dir == "/data/logs"
    fv.log_dir = "/other"
    assert fv.flag_values_dict() == {"log_dir": "/other"}


def test_paths(monkeypatch):
    monkeypatch.delenv("DTF_LOGS_ROOT", raising=False)
    monkeypatch.delenv("DTF_DATA_ROOT", raising=False)
    assert paths.get_logs_path(root="/a/logs") == "/a/logs"
    assert paths.get_data_path(dataset_name="u/ds", local_root="/a/data", local_repo="repo", path="p") == "/a/data/repo/p"
    monkeypatch.setenv("DTF_LOGS_ROOT", "/logs")
    monkeypatch.setenv("DTF_DATA_ROOT", "/data")
    assert paths.get_logs_path(root="/a/logs") == "/logs"
    assert paths.get_data_path(dataset_name="u/ds", local_root="/a", local_repo="r", path="p") == "/data/u/ds/p"
