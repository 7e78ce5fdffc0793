"""rocprofv3 command builder, counter-pass splitting, CSV summarisers and the step timer (all host-side)."""
import csv
import time

import pytest

from distributed_tensorflow_amd.utils import (StepTimer, kernel_time_by_name, pivot_counters, rocprof_command,
                                              split_counter_passes)


def test_counter_passes_respect_slot_limits():
    sq = ["SQ_WAVES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES",
          "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_SALU", "SQ_ACTIVE_INST_ANY"]
    passes = split_counter_passes(sq)
    assert [len(p) for p in passes] == [8, 1]
    # FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2: they cannot share a pass
    passes = split_counter_passes(["TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum", "TCC_FETCH_SIZE_sum", "TCC_WRITE_SIZE_sum"])
    assert passes == [["TCC_EA0_RDREQ_sum", "TCC_BUBBLE_sum"], ["TCC_FETCH_SIZE_sum"], ["TCC_WRITE_SIZE_sum"]] or \
        all("TCC_FETCH_SIZE_sum" not in p or "TCC_WRITE_SIZE_sum" not in p for p in passes)
    # _sum / _avr of one counter count once
    assert len(split_counter_passes(["TA_BUSY_avr", "TA_BUSY_sum", "TA_TA_BUSY_sum"])) == 1
    with pytest.raises(ValueError):
        split_counter_passes(["XYZ_SOMETHING"])


def test_rocprof_command_shapes():
    cmd = rocprof_command(["python3", "bench.py", "--steps", "5"], "out", "mlp")
    assert cmd[:3] == ["rocprofv3", "--kernel-trace", "--stats"] and cmd[-4:] == ["python3", "bench.py", "--steps", "5"]
    assert cmd[cmd.index("--") + 1] == "python3"
    pmc = rocprof_command(["./bench"], "out", counters=["SQ_WAVES", "GRBM_GUI_ACTIVE"])
    assert "--pmc" in pmc and "--stats" not in pmc
    with pytest.raises(ValueError):
        rocprof_command(["./bench"], "out", counters=["SQ_WAVES"], extra=["--sys-trace"])
    with pytest.raises(ValueError):
        rocprof_command(["env", "X=1", "./bench"], "out")
    with pytest.raises(ValueError):
        rocprof_command(["./bench"], "out", counters=["SQ_A%d" % i for i in range(9)])


def test_csv_summaries(tmp_path):
    kt = tmp_path / "k_kernel_trace.csv"
    with open(kt, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        w.writerow(["KERNEL_DISPATCH", "gemm", 1000, 51000])
        w.writerow(["KERNEL_DISPATCH", "adam", 52000, 57000])
        w.writerow(["KERNEL_DISPATCH", "gemm", 60000, 100000])
    agg = kernel_time_by_name(str(kt))
    assert list(agg) == ["gemm", "adam"] and agg["gemm"]["calls"] == 2 and agg["gemm"]["total_us"] == pytest.approx(90.0)
    cc = tmp_path / "c_counter_collection.csv"
    with open(cc, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        w.writerow([1, "gemm", "SQ_WAVES", 10])
        w.writerow([1, "gemm", "SQ_WAVES", 6])
        w.writerow([1, "gemm", "GRBM_GUI_ACTIVE", 100])
        w.writerow([2, "adam", "SQ_WAVES", 3])
    rows = pivot_counters(str(cc))
    assert rows[0] == {"name": "gemm", "SQ_WAVES": 16.0, "GRBM_GUI_ACTIVE": 100.0} and rows[1]["SQ_WAVES"] == 3.0


def test_step_timer():
    t = StepTimer(every=3, batch_size=50)
    assert t.step() is None and t.step() is None
    time.sleep(0.01)
    out = t.step()
    assert out is not None and out["examples/sec"] == pytest.approx(out["global_step/sec"] * 50)
    assert out["global_step/sec"] < 3 / 0.01 * 1.01
