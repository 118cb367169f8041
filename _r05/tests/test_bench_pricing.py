"""bench.py's roofline pricing (host logic, no GPU): SURVEY.md 8(d)'s per-convolution formula on the canonical scene's own
numbers, the kernel naming that has to equal rocprofv3's, and the traffic look-up by exact instance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_conv_bytes_follow_survey_8d_on_the_canonical_level0():
    # SURVEY 8(d), level 0 of the canonical scene: M = 142,920, P27 = 1,332,424, C = 16 -> one k3 16->16 convolution:
    # 2*(M*16 + M*16) + 2*27*16*16 + 8*P27 bytes, 2*P27*16*16 flops
    M, P = 142920, 1332424
    b, f = bench.conv_bytes_8d([M, M, 27, 16, 16], {M: P})
    assert b == 2.0 * (M * 16 + M * 16) + 2.0 * 27 * 16 * 16 + 8.0 * P
    assert f == 2.0 * P * 16 * 16
    # eleven such convolutions = SURVEY's L0 row (194.0 MB, 6.579 GFLOP) up to its 1x1 / concat layers
    assert abs(11 * b / 1e6 - 218.0) < 1.0 and abs(11 * f / 1e9 - 7.50) < 0.01
    # stride-2 (K = 8): one pair per fine row; 1x1: one pair per output row
    b, f = bench.conv_bytes_8d([M, 35127, 8, 16, 32], {})
    assert b == 2.0 * (M * 16 + 35127 * 32) + 2.0 * 8 * 16 * 32 + 8.0 * M and f == 2.0 * M * 16 * 32
    b, f = bench.conv_bytes_8d([M, M, 1, 32, 16], {})
    assert b == 2.0 * (M * 32 + M * 16) + 2.0 * 32 * 16 + 8.0 * M


def test_kernel_names_are_rocprofv3_names():
    tags = [0, 0, 27, 16, 16, 1, 1, 1, 4, 0, 27, 2]
    assert bench.prof_kernel_name(0, tags) == "spconv_fwd2_kernel<1, true, true, 4, false, 27, 2, false>"
    tags[11] = 1002                                         # the T16 template flag travels in the ST tag
    assert bench.prof_kernel_name(0, tags) == "spconv_fwd2_kernel<1, true, true, 4, false, 27, 2, true>"
    # round 5: the other entry points of the same kernel body (+ 4000 offset compaction, + 2000 offset split), one family
    assert bench.prof_kernel_name(0, [0, 0, 27, 136, 16, 1, 1, 1, 16, 0, 27, 5017]) == "spconv_fwd2_c_kernel<1, 16, 17, true>"
    assert bench.prof_kernel_name(0, [0, 0, 27, 136, 16, 1, 1, 1, 16, 0, 27, 4017]) == "spconv_fwd2_c_kernel<1, 16, 17, false>"
    assert bench.prof_kernel_name(0, [0, 0, 27, 136, 16, 1, 1, 1, 16, 0, 27, 3017]) == "spconv_fwd2_ks_kernel<17, true, 4>"
    assert bench.kernel_family("spconv_fwd2_c_kernel<1, 16, 17, true>") == "spconv_fwd2_kernel" == bench.kernel_family("void spconv_fwd2_ks_kernel<17, true, 4>")
    assert bench.kernel_family("spconv_fwd2_split_kernel<4, true, false>") == "spconv_fwd2_split_kernel" and bench.kernel_family("cl_bfs2_kernel") == "cl_bfs2_kernel"
    assert bench.prof_kernel_name(2, [0, 0, 27, 64, 64, 4, 1, 0, -1, 0, 0, 0]) == "spconv_fwd2_split_kernel<4, true, false>"
    assert bench.prof_kernel_name(3, [4096, 128, 128, 1, 0, 0, 0, 0, 0, 0, 0, 0]) == "hg_gemm_tiled_kernel"
    assert bench.prof_kernel_name(3, [32, 512, 512, 1, 1, 1, 16, 0, 0, 0, 0, 0]) == "hg_gemm_kernel<1, true, 16>"
    assert bench.prof_kernel_name(5, [0] * 12) == "cl_bfs2_kernel"


def test_roofline_object_prices_gemm_against_fp32_mfma_and_looks_traffic_up_by_instance():
    r = dict(family=3, launches_sampled=10, launches_per_step=5.0, avg_launch_us=20.0, ms_per_step=0.1, algorithmic_bytes_per_launch=1e6,
             bytes_moved_by_design_per_launch=1e6, flops_per_launch=2e9, achieved_gbs=50.0, achieved_tflops=100.0)
    o = bench.roofline_object("hg_gemm_tiled_kernel", r, {"hg_gemm_tiled_kernel": {"hbm_bytes_per_launch": 3e6}}, 13)
    assert o["bound"] == "mfma" and o["peak"] == bench.MFMA_F32_PEAK_TFLOPS and abs(o["frac"] - 100.0 / 157.3) < 1e-9
    assert o["traffic"] == 3e6 and abs(o["traffic_over_algorithmic"] - 3.0) < 1e-12
    c = dict(r, family=0)
    o = bench.roofline_object("spconv_fwd2_kernel<1, true, true, 4, false, 27, 2, true>", c, {"spconv_fwd2_kernel": {"hbm_bytes_per_launch": 9e6}}, 13)
    assert o["bound"] == "hbm" and o["traffic"] is None          # a family average is never quoted next to one instance
