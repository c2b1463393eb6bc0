"""Host-side checks of bench.py's reporting: the result line stays short enough for a reader that keeps
only the tail of stdout, the roofline classification follows the counters, the byte model matches the
closed-form step totals."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _row(name, bound="hbm", unit="GB/s"):
    return {"kernel": name, "avg_launch_us": 281.12, "launches_per_step": 3, "ms_per_step": 0.8434, "bound": bound,
            "achieved": 4834.2, "peak": 8000.0, "unit": unit, "frac": 0.6043, "per_use_bytes_per_launch": 1359000000,
            "compulsory_bytes_per_launch": 697000000, "compulsory_frac_of_hbm_peak": 0.31, "traffic": 1585445643,
            "traffic_over_per_use": 1.167, "traffic_rate_gbs": 5640.1}


def _leg(name, n, e, shape):
    rows = [_row(f"{name}_kernel_with_a_long_template_name<4,{i}>") for i in range(40)]
    kernels = {r["kernel"]: {"launches": 60, "avg_us": 281.1, "total_ms_per_step": 0.8434} for r in rows}
    return {"ms_per_step": 6.12234, "edges_per_s": 333211234.5, "roofline": dict(rows[0], profiled_ms_per_step=7.1),
            "roofline_next": rows[1:4], "roofline_rows": rows, "kernels": kernels, "batch_assembly_ms": 0.6312,
            "end_to_end_ms": 6.694, "eager_ms": 6.2793, "adam_ms": 0.0241, "n": n, "e": e, "shape": shape,
            "kernel_sum_ms": 6.2,
            "workload": "StreamSpot batch-level training, batch of 12 graphs per GPU (streamspot_batch): N=96000 "
                        "E=2040000 n_dim=8 e_dim=20 hidden=256 layers=4; one-hot attributes + compute_loss fwd + bwd "
                        "+ Adam, dropout on, mask perms and sampled pairs pre-generated on the device",
            "dims": shape}


def test_result_line_fits_a_4kb_tail(tmp_path):
    args = argparse.Namespace(steps=20, warmup=5, rows_file=str(tmp_path / "rows.json"))
    leg = _leg("top", 96000, 2040000, (8, 20, 256, 4))
    entity = _leg("entity", 700000, 3000000, (5, 17, 64, 3))
    cpu = {"value": 186736.0, "unit": "edges/s", "cores": 16, "kind": "port", "cpu": "AMD EPYC; " + "x" * 120,
           "sample": "torch-CPU restatement of the DGL path (oracle/gat_oracle.py), same recipe on the full batch of "
                     "12 StreamSpot-sized graphs (N=96000, E=2040000), as the GPU step, 2 timed step(s) after one "
                     "warm-up step, 16 threads"}
    knn = {"metric": "KNN-score queries/sec", "value": 662530.4, "unit": "queries/s", "ms_per_step": 1032.567,
           "dtype": "f64", "steps": 2,
           "config": {"short": "E3-Trace-shaped scoring: train 2604530x64 over 1 rank(s), 684107 test rows + 50000-row "
                               "mean_distance search per step, k=10", "workload": "w" * 300},
           "roofline": dict(_row("knn_filter_bf16x6_mfma", "mfma", "TFLOP/s"), arithmetic="a" * 200),
           "kernels": {f"knn_kernel_{i}": {"launches": 6, "total_ms_per_step": 1003.554} for i in range(8)},
           "cadets": {"k": 200, "value": 123456.7, "unit": "queries/s", "ms_per_step": 2890.123,
                      "roofline": bench._roof(_row("knn_filter_bf16x6_mfma", "mfma", "TFLOP/s"))},
           "cpu_baseline": {"value": 2167.1, "unit": "queries/s", "cores": 16, "kind": "reference", "cpu": "c" * 130,
                            "sample": "sklearn.neighbors.NearestNeighbors(n_jobs=16).kneighbors: 2000 queries against "
                                      "the full 2604530 x 64 train set, k=10"}}
    configs = {k: {"n": 126356, "e": 351388, "ms_per_step": 6.21234, "eager_ms": 5.91234, "edges_per_s": 5.6e7}
               for k in ("c1", "c3_avg", "c3_max")}
    rows_path = bench.write_rows_file(args, leg, entity, knn)
    line = json.dumps(bench.compact_line(args, 1, leg, entity, knn, configs, cpu, rows_path))
    assert len(line) < 4096, len(line)
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    for roof in (out["roofline"], out["entity_level"]["roofline"], out["knn"]["roofline"]):
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(roof)
    assert {"value", "unit", "cores", "kind", "sample"} <= set(out["cpu_baseline"])
    assert set(out["configs"]) == {"c1", "c3_avg", "c3_max"}
    # the long tables went to the side file
    doc = json.load(open(tmp_path / "rows.json"))
    assert len(doc["top"]["roofline_rows"]) == 40 and len(doc["entity_level"]["kernels"]) == 40


def test_sweep_classification_follows_the_counters():
    # THEIA-shaped source sweep: the fabric moves more than the per-use count -> HBM-bound, priced per use
    n, e, h, c = 700000, 3000000, 4, 64
    per_use, comp, gath, table = bench.sweep_bytes("bwd_src", n, e, e, h, c, None)
    assert per_use == e * (8 + 16 * h + 4 * c) + n * (8 + 4 * h + 8 * c)
    r = bench.classify_sweep(per_use, comp, gath, table, 1585445643, 281.1e-6)
    assert r["bound"] == "hbm" and abs(r["frac"] - per_use / 281.1e-6 / 8e12) < 1e-3 and 0.59 < r["frac"] < 0.62
    # hidden-256 batch: a third of the per-use bytes cross the fabric -> rows come from L2, no fraction above 1
    n, e, c = 96000, 2040000, 256
    per_use, comp, gath, table = bench.sweep_bytes("bwd_src", n, e, e, h, c, None)
    r = bench.classify_sweep(per_use, comp, gath, table, 882649818, 230e-6)
    assert r["bound"] == "l2-gather" and r["frac"] < 1.0 and r["peak"] == bench.L2_GATHER_PEAK_GBS
    # first layer on code words: no gathered node rows in forward, the source sweep keeps its gradient-row gather
    n, e, c = 700000, 3000000, 64
    per_use, comp, gath, table = bench.sweep_bytes("fwd_coded", n, e, e, h, c, "linear")
    assert per_use == e * 12 + n * (12 + 8 * h + 4 * c) and gath == 0 and comp <= per_use
    assert bench.classify_sweep(per_use, comp, gath, table, 371709291, 134e-6)["bound"] == "hbm"
    dense = bench.sweep_bytes("bwd_src", n, e, e, h, c, None)
    coded = bench.sweep_bytes("bwd_src_coded", n, e, e, h, c, None)
    assert coded[2] == dense[2] and dense[0] - coded[0] == n * (4 * h + 4 * c - 4)
    n, e, c = 96000, 2040000, 256
    per_use, comp, gath, table = bench.sweep_bytes("bwd_src", n, e, e, h, c, None)
    # no counters: the per-use rate decides
    assert bench.classify_sweep(per_use, comp, gath, table, None, 230e-6)["bound"] == "l2-gather"
    assert bench.classify_sweep(per_use, comp, gath, table, None, 400e-6)["bound"] == "hbm"


def test_step_byte_model_matches_the_closed_form_totals():
    # (N, E, n_dim, hidden, layers) -> GB per step, dense GFLOP per step
    for shape, gb, gf in (((700000, 3000000, 5, 64, 3), 14.6, 91.7), ((8000, 170000, 8, 256, 4), 2.81, 72.5),
                          ((96000, 2040000, 8, 256, 4), 30.9, 329.5), ((36812, 96018, 8, 256, 4), 3.18, 165.1)):
        by, fl = bench.step_algorithmic(*shape)
        assert abs(by / 1e9 - gb) / gb < 0.01, (shape, by)
        assert abs(fl / 1e9 - gf) / gf < 0.02, (shape, fl)


def test_projection_launch_shapes():
    assert bench.dense_launch("proj_fwd<192,64>", 700000, 64, 3, 22500) == (2.0 * 700000 * 192 * 64,
                                                                            4.0 * 700000 * 256, 700000)
    assert bench.dense_launch("proj_bwd_weight<64,384>", 700000, 64, 3, 22500)[2] == 22500
    assert bench.dense_launch("sce_fwd_bwd", 1, 64, 3, 1) is None
