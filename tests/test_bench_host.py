"""bench.py's host-side arithmetic, no GPU: where `roofline.traffic` and `roofline.pipes.valu_insts_per_block` come from
(VERDICT r04 #8a / ADVICE r04: the VALU-per-block figure was a constant copied by hand and went stale with the next
kernel change -- it is now derived from the SQ_INSTS_VALU pass of the same run, or quoted from profiles/pmc_valu.json
with its provenance when no pass can run)."""
import json
import os

import bench

GIB = 1 << 30


def test_valu_per_block_is_derived_from_the_runs_own_counter_pass():
    # 2.237e8 wave instructions per 1 GiB step (profiles/r04_ctr_rocprof_summary.txt) -> 213 per block
    v, src = bench.valu_per_block_of({"SQ_INSTS_VALU": 2.237e8}, GIB, "ctr")
    assert abs(v - 213.3) < 0.2 and src.startswith("measured in this run")
    v2, _ = bench.valu_per_block_of({"SQ_INSTS_VALU": 2.0e8}, GIB, "ctr")       # a different kernel -> a different figure
    assert abs(v2 - 190.7) < 0.2


def test_valu_per_block_falls_back_to_the_recorded_figure_and_says_so(tmp_path):
    with open(os.path.join(bench.ROOT, "profiles", "pmc_valu.json")) as f:
        rec = json.load(f)
    for wl in ("ctr", "gcm", "xts"):
        v, src = bench.valu_per_block_of({}, GIB, wl)
        assert v == rec[wl]["valu_insts_per_block"] and src.startswith("recorded in profiles/") and "not measured in this run" in src
        assert os.path.exists(os.path.join(bench.ROOT, rec[wl]["source"]))
    assert bench.valu_per_block_of({}, GIB // 2, "ctr") == (None, None)            # another size: no figure rather than a stale one
    assert bench.valu_per_block_of({}, GIB, "ecb") == (None, None)
    (tmp_path / "pmc_valu.json").write_text(json.dumps({"ctr": {"bytes_per_gpu": GIB, "valu_insts_per_block": 7,
                                                                "source": "x", "build": "y"}}))
    assert bench.valu_per_block_of({}, GIB, "ctr", str(tmp_path))[0] == 7


def test_traffic_from_the_two_counter_passes():
    t, src = bench.traffic_of({"FETCH_SIZE": 527966.0, "WRITE_SIZE": 1059605.0})
    assert t == int((2 * 527966 + 1059605) * 1024) and src.startswith("measured in this run")
    assert bench.traffic_of({"FETCH_SIZE": 1.0}) == (None, None)
