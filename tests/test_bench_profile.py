"""The committed counter profile must describe the kernels of THIS tree (a host-only edit staled profiles/r3_pmc.json and the
driver's bench line lost its roofline fraction). CPU-only: no GPU is touched, nothing under oracle/ is used."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_device_sources_are_the_kernel_translation_unit_and_its_includes():
    src = bench.device_sources()
    assert "frt_kernels.hip" in src and "frt_trace.hpp" in src and "frt_shade.hpp" in src and "frt_math.hpp" in src
    # host-only code and the experiments build do not decide the product's device code
    for f in ("frt_bvh_opt.hpp", "frt_scene.hpp", "frt_renderer.hip", "frt_multi.hip", "frt_loader.hpp"):
        assert f not in src, f
    assert not any(f.startswith("experiments") for f in src)


def test_committed_profile_matches_head():
    """Fails the moment a commit changes device code without re-running tools/profile_all.sh + tools/pmc_to_json.py."""
    pmc = bench.load_pmc()
    assert pmc is not None, f"{bench.PMC_JSON} is missing or unreadable"
    assert pmc["stale"] is False, (f"{os.path.relpath(bench.PMC_JSON, ROOT)} was measured on other device code: source hash {pmc.get('source_hash')} vs "
                                    f"{bench.source_hash()}, code objects {pmc.get('code_object_hash')} vs {bench.code_object_hash()}")
    for k in ("valu_insts_per_frame", "hbm_bytes_per_frame", "cycles_per_valu_inst", "shader_clock_ghz", "kernels"):
        assert k in pmc, k
    assert pmc["valu_insts_per_frame"] > 1e8 and pmc["hbm_bytes_per_frame"] > 1e8


def test_code_object_hash_reads_the_fatbin_of_the_built_library():
    lib = os.path.join(ROOT, "fast-raytracing-wgpu_amd", "lib", "libfrt.so")
    if not os.path.exists(lib):
        import pytest
        pytest.skip("library not built")
    h = bench.code_object_hash()
    assert isinstance(h, str) and len(h) == 16
    assert bench.code_object_hash(__file__) is None          # not an ELF file


def test_roofline_block_carries_numbers_from_the_profile():
    stages = {n: {"algorithmic_bytes": 1.0e9, "event_ms": 0.3, "rays": 1e6, "bytes_per_ray": 700.0} for n in bench.STAGES}
    roof = bench.roofline_block(1, 1.5, stages, 1920 * 1080)
    pmc = json.load(open(bench.PMC_JSON))
    for k in ("achieved", "frac", "traffic"):
        assert isinstance(roof[k], float) and roof[k] > 0, k
    assert abs(roof["frac"] - pmc["valu_insts_per_frame"] * 64.0 / 1.5e-3 / 1e12 / roof["peak"]) < 1e-9
    assert roof["hbm_actual"]["frac"] > 0 and roof["valu_issue"]["frac_at_measured_clock"] > 0 and roof["pmc_source"]["stale"] is False
