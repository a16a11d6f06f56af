"""CPU: bench.py's reading of the committed PMC summaries under profiles/ can never take the JSON line down.

Round 3's driver run died with KeyError 'families' after the timed loop: a glob matched the summary of ANOTHER model whose
schema differed.  These tests run the enrichment code on the committed profiles/ for every --model, on broken side files,
and pin the keys every usable committed summary must carry."""
import glob
import json
import os
import shutil

import pytest

from _util import ROOT

import bench

MODELS = ("quadtree", "attention", "cnn_lstm", "quadtree3d")


@pytest.mark.parametrize("model", MODELS)
def test_pmc_enrichment_never_raises_on_the_committed_profiles(model):
    for kind in ("traffic", "mfma_busy"):
        rec, src = bench.pmc_summary(kind, model)
        assert isinstance(src, str) and src
        if rec is not None:
            assert bench.summary_problems(rec, kind) == []
            assert bench.summary_identity(rec, os.path.join(ROOT, src))[:2] == (kind, model)
    fams = bench.ROOFLINE_FAMILIES_3D if model == "quadtree3d" else bench.ROOFLINE_FAMILIES
    traffic, src = bench.roofline_traffic(model, 256, "bf16", fams)
    assert traffic is None or traffic > 0
    assert isinstance(src, str)
    hbm = bench.guarded("hbm", bench.step_hbm, model, 256, "bf16", 6e-3)
    assert isinstance(hbm, dict) and "error" not in hbm
    for mode in ("train", "eval"):
        busy = bench.guarded("busy", bench.step_mfma_busy, model, 256, "bf16", mode, 6e-3)
        assert isinstance(busy, dict) and "error" not in busy


def test_every_committed_summary_on_the_current_sources_has_the_keys_the_reader_uses():
    """A summary whose kernel_sources_sha1 is this tree's is one bench.py may pick: it must satisfy the schema of its kind
    (a summary of another shape has to live under another name / kind, not be skipped silently forever)."""
    cur = bench.kernel_sources_sha1()
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*.json"))):
        try:
            rec = json.load(open(f))
        except Exception:
            continue
        if not isinstance(rec, dict) or rec.get("kernel_sources_sha1") != cur:
            continue
        kind = bench.summary_identity(rec, f)[0]
        if kind in bench.SUMMARY_KEYS:
            assert bench.summary_problems(rec, kind) == [], f


def test_summaries_of_other_models_or_shapes_are_never_picked(tmp_path, monkeypatch):
    """The round-3 crash, restated: a quadtree3d summary (same sources, other keys), a truncated file, a summary with a
    families entry of the wrong shape and a headline summary of another batch all sit next to the right one."""
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_sources_sha1", lambda: "abc")
    good = {"kind": "traffic", "model": "quadtree", "batch": 256, "dtype": "bf16", "kernel_sources_sha1": "abc",
            "steps_profiled": 5,
            "families": {"conv_pt_kernel": {"hbm_bytes_per_launch": 100, "launches_profiled": 10},
                         "bn_act": {"hbm_bytes_per_launch": 50, "launches_profiled": 20}}}
    (prof / "r03_traffic.json").write_text(json.dumps(good))
    # newer by name, same sources: the legacy 3-D layout that crashed round 3
    (prof / "r03_quadtree3d_traffic.json").write_text(json.dumps(
        {"command": "bench.py --model quadtree3d --steps 3", "kernel_sources_sha1": "abc", "hbm_gb_per_step": {"total": 21.4},
         "by_kernel": {}}))
    (prof / "r04_traffic.json").write_text("{ truncated")
    (prof / "r05_traffic.json").write_text(json.dumps(dict(good, families={"conv_pt_kernel": {"avg_us": 3}})))
    (prof / "r06_traffic.json").write_text(json.dumps(dict(good, batch=8)))
    (prof / "r07_traffic.json").write_text(json.dumps([1, 2, 3]))
    rec, src = bench.pmc_summary("traffic")
    assert rec == good and src == os.path.join("profiles", "r03_traffic.json")
    assert bench.roofline_traffic("quadtree", 256, "bf16", bench.ROOFLINE_FAMILIES)[0] == 100
    assert bench.step_hbm("quadtree", 256, "bf16", 1e-3)["bytes_per_step"] == (100 * 10 + 50 * 20) // 5
    rec, why = bench.pmc_summary("traffic", "quadtree3d")
    assert rec is None and "families" in why
    assert bench.roofline_traffic("quadtree3d", 256, "bf16", bench.ROOFLINE_FAMILIES_3D)[0] is None
    assert bench.step_hbm("quadtree3d", 256, "bf16", 1e-3)["bytes_per_step"] is None
    rec, why = bench.pmc_summary("mfma_busy")
    assert rec is None and "no committed mfma_busy summary" in why
    # a profiles/ that is not even a directory
    shutil.rmtree(prof)
    prof.write_text("not a directory")
    assert bench.pmc_summary("traffic")[0] is None


def test_guarded_turns_an_exception_into_a_field():
    def boom():
        raise KeyError("families")
    out = bench.guarded("hbm summary", boom)
    assert "families" in out["error"]
