"""CPU: kernel name -> hot-path family (estdepth_amd/profiling.py), on the names rocprofv3 reports for the current kernels; a family
that falls through silently would put the eager brackets back into the bench line (tests/test_gpu_bench_contract.py)."""
from estdepth_amd.profiling import family_of

NAMES = {
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 1, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 2, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, true, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:33->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, false, true>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->16",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, true, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, false, true, false, true>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->16",   # reset-gated
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, false, false, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino2_kernel<8, 0, true, false, true, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:33->33",
    "void (anonymous namespace)::conv3d_xout_kernel(estd_conv3d_desc, int, int, int)": "conv3d:33->1",
    "void (anonymous namespace)::conv3d_wino2x_kernel<0, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino3_kernel<0, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",     # three-axis kernel: <read-back kind, GroupNorm, scalar channel>
    "void (anonymous namespace)::conv3d_wino3_kernel<2, false, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino3_kernel<0, true, false>(estd_conv3d_desc, int, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::conv3d_wino3_kernel<0, false, true>(estd_conv3d_desc, int, int, int, int)": "conv3d:33->32",
    "(anonymous namespace)::conv3d_wino2_c16_kernel(estd_conv3d_desc, int, int, int, int)": "conv3d:16->16",
    "void (anonymous namespace)::conv3d_wino_kernel<true, true>(estd_conv3d_desc, int, int, int, int)": "conv3d:33->33",
    "void (anonymous namespace)::conv3d_k3_kernel<16, 1, false, false>(estd_conv3d_desc, int, int, int)": "conv3d:16->16",
    "void (anonymous namespace)::conv3d_k3_kernel<32, 2, false, false>(estd_conv3d_desc, int, int, int)": "conv3d:32->32",
    "void (anonymous namespace)::warp_attention_kernel<3>(float const*, (anonymous namespace)::WarpAttnArgs, float const*)": "warp_attention",
    "void (anonymous namespace)::homo_warp_costvol_kernel<true>(float const*, float const*)": "homo_warp_costvol",
    "(anonymous namespace)::softargmin_up_kernel(float const*, float const*, float*, float*, int, int, int, int, int)": "softargmin",
    "(anonymous namespace)::gru_reset_kernel(float const*)": "gru_elementwise",
    "(anonymous namespace)::gru_blend_kernel(float const*)": "gru_elementwise",
    "(anonymous namespace)::conv2d_wino2_kernel(estd_conv2d_desc, int, int, int)": None,
    "void (anonymous namespace)::conv1x1_nhwc_kernel<4, 4, 2>(estd_conv1x1_desc, int, int, int, int)": None,
    "void (anonymous namespace)::conv1x1_lds_kernel<64, 64, 2, 2, 1, 4>(estd_conv1x1_desc, int, int, int, int, int)": None,
    "Cijk_Ailk_Bljk_S_B_Bias_HA_S_SAV_UserArgs_MT128x128x16": None,
}


def test_family_of_current_kernel_names():
    for name, fam in NAMES.items():
        assert family_of(name) == fam, name
