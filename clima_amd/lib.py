"""Loader for the HIP C-ABI library (include/clima_radtran_hip.h).

There is no fallback: if the shared library is missing or does not load, importing the
product path raises.  Nothing here touches oracle/.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLIMA_HIP_LIB: another build of the same library (A/B timing of kernel variants on one box)
LIB_PATH = os.environ.get("CLIMA_HIP_LIB") or os.path.join(_HERE, "csrc", "libclima_radtran_hip.so")
ERR_LEN = 1024

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_bp = C.POINTER(C.c_bool)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
_err = C.c_char_p

# name -> argtypes, exactly as declared in include/clima_radtran_hip.h
SIGNATURES = {
    "allocate_radtran": [_vpp],
    "deallocate_radtran": [_vp],
    "radtran_create_begin": [_vp, _ip, _ip, _ip, _ip, _dp, _err],
    "radtran_add_ktable": [_vp, _ip, _ip, _dp, _ip, _dp, _ip, _dp, _dp, _err],
    "radtran_add_xsection": [_vp, _ip, _ip, _ip, _ip, _ip, _dp, _dp, _err],
    "radtran_set_water_continuum": [_vp, _ip, _ip, _dp, _dp, _dp, _err],
    "radtran_add_particle": [_vp, _ip, _ip, _dp, _dp, _dp, _dp, _err],
    "radtran_set_channels": [_vp, _ip, _dp, _ip, _dp, _err],
    "radtran_set_photons_sol": [_vp, _ip, _dp, _err],
    "radtran_create_end": [_vp, _ip, _dp, _err],
    "radtran_radiate_wrapper": [_vp, _dp, _ip, _dp, _ip, _dp, _ip, _ip, _dp, _ip, _dp, _ip, _ip, _ip, _dp, _ip, _ip,
                                _dp, _ip, _ip, _err],
    "radtran_toa_fluxes_wrapper": [_vp, _dp, _ip, _dp, _ip, _dp, _ip, _ip, _dp, _ip, _dp, _ip, _ip, _ip, _dp, _ip, _ip,
                                   _dp, _ip, _ip, _dp, _dp, _err],
    "radtran_apply_radiation_enhancement": [_vp, _dp],
    "radtran_set_custom_optical_properties": [_vp, _ip, _dp, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp, _ip, _ip, _dp, _err],
    "radtran_unset_custom_optical_properties": [_vp],
    "radtran_opacities2yaml_wrapper_1": [_vp, _ip, _vpp],
    "radtran_opacities2yaml_wrapper_2": [_vp, _vpp, _ip, _err],
    "radtran_set_names": [_vp, _err, _err, _err],
    "radtran_set_opacity_labels": [_vp, _err, _err, _err, _err],
    "radtran_fused_set": [_vp, _ip],
    "radtran_fused_get": [_vp, _ip],
    "radtran_fused_fallbacks_get": [_vp, _ip],
    "radtran_ir_green_set": [_vp, _ip],
    "radtran_ir_green_get": [_vp, _ip, _ip],
    "radtran_coop_items_set": [_vp, _ip],
    "radtran_coop_items_get": [_vp, _ip],
    "radtran_fused_spins_set": [_vp, _ip],
    "radtran_fused_spins_get": [_vp, _ip],
    "radtran_spectra_get_all": [_vp, _bp, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _err],
    "radtran_spectra_release": [_vp],
    "radtran_create_from_files": [_vp, C.c_char_p, C.c_char_p, _ip, _dp, _ip, C.c_char_p, _err],
    "radtran_load_from_files": [_vp, C.c_char_p, C.c_char_p, _ip, C.c_char_p, _err],
    "clima_test_host_tables_digest": [_vp, C.POINTER(C.c_ulonglong)],   # 8 digests
    "radtran_dims_get": [_vp, _ip, _ip, _ip, _ip, _ip],
    "radtran_names_get": [_vp, _ip, C.c_char_p, C.c_char_p],
    "radtran_toa_fluxes_batch": [_vp, _ip, _dp, _dp, _dp, _dp, _dp, _ip, _dp, _dp, _dp, _dp, _dp, _err],
    "radtran_radiate_ir_batch": [_vp, _ip, _dp, _ip, _ip, _dp, _dp, _dp, _dp, _err],
    "radtran_upload_column": [_vp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _err],
    "radtran_radiate_resident": [_vp, _ip, _ip, _err],
    "radtran_synchronize": [_vp, _err],
    "radtran_flux_device_ptr": [_vp, _vpp, _ip],
    "radtran_set_bin_shard": [_vp, _ip, _ip, _err],
    "radtran_bin_shard_get": [_vp, _ip, _ip, _ip, _ip, _ip, _ip],
    "radtran_finish_reduced": [_vp, _err],
    "radtran_set_device": [_ip, _err],
    "radtran_comm_unique_id": [_err, _err],
    "radtran_comm_init_rank": [_vp, _ip, _ip, _err, _err],
    "radtran_comm_init_file": [_vp, _ip, _ip, _err, _err],
    "radtran_comm_get": [_vp, _ip, _ip, _ip],
    "radtran_comm_destroy": [_vp],
    "radtran_stream_get": [_vp, _vpp],
    "radtran_profile_set": [_vp, _ip],
    "radtran_kernel_time_get": [_vp, _ip, _dp, _ip, _err],
    "radtran_profile_reset": [_vp],
    "radtran_profile_stride_set": [_vp, _ip],
    "radtran_algorithmic_bytes": [_vp, _dp, _dp, _dp, _dp, _err],
    "radtran_algorithmic_nodes": [_vp, _dp, _dp, _dp, _dp, _err],
    "radtran_opr_get": [_vp, _dp, _dp, _dp, _dp, _err],
    "clima_bench_toa_fluxes": [_vp, _ip, _dp, _ip, _dp, _ip, _dp, _ip, _ip, _dp, _ip, _dp, _ip, _ip, _ip, _dp, _ip, _ip,
                               _dp, _dp, _dp, _dp, _err],
    "clima_bench_resident_sync": [_vp, _ip, _dp, _err],
    "clima_bench_resident_graph": [_vp, _ip, _ip, _dp, _err],
    "clima_test_device_exp": [_ip, _dp, _dp, _err],
    "clima_test_device_exp_table": [_ip, _ip, _dp, _dp, _err],
    "clima_test_device_rcp": [_ip, _dp, _dp, _err],
    "clima_test_wave_scan": [_ip, _dp, _dp, _dp, _err],
    "clima_test_two_stream": [_ip, _ip, _ip, _ip, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _err],
    "clima_test_green_far_form_set": [_ip],
    "radtran_batch_pin_results_set": [_vp, _ip],
    "radtran_batch_pin_results_get": [_vp, _ip],
    "clima_test_ir_response": [_ip, _ip, _dp, _dp, _dp, _dp, _dp, _ip, _ip, _dp, _dp, _dp, _err],
    "radtran_set_bolometric_flux_wrapper": [_vp, _dp],
    "radtran_bolometric_flux_wrapper": [_vp, _dp],
    "radtran_skin_temperature_wrapper": [_vp, _dp, _dp],
    "radtran_equilibrium_temperature_wrapper": [_vp, _dp, _dp],
    "radtran_zenith_u_get_size": [_vp, _ip],
    "radtran_zenith_u_get": [_vp, _ip, _dp],
    "radtran_zenith_u_set": [_vp, _ip, _dp],
    "radtran_zenith_weights_get": [_vp, _ip, _dp],
    "radtran_zenith_weights_set": [_vp, _ip, _dp],
    "radtran_surface_albedo_get_size": [_vp, _ip],
    "radtran_surface_albedo_get": [_vp, _ip, _dp],
    "radtran_surface_albedo_set": [_vp, _ip, _dp],
    "radtran_surface_emissivity_get_size": [_vp, _ip],
    "radtran_surface_emissivity_get": [_vp, _ip, _dp],
    "radtran_surface_emissivity_set": [_vp, _ip, _dp],
    "radtran_has_hard_surface_get": [_vp, _bp],
    "radtran_has_hard_surface_set": [_vp, _bp],
    "radtran_photon_scale_factor_get": [_vp, _dp],
    "radtran_photon_scale_factor_set": [_vp, _dp],
    "radtran_ir_tau_min_get": [_vp, _dp],
    "radtran_ir_tau_min_set": [_vp, _dp],
    "radtran_diurnal_fac_get": [_vp, _dp],
    "radtran_diurnal_fac_set": [_vp, _dp],
    "radtran_ir_get": [_vp, _vpp],
    "radtran_sol_get": [_vp, _vpp],
    "radtran_wrk_ir_get": [_vp, _vpp],
    "radtran_wrk_sol_get": [_vp, _vpp],
    "radtran_f_total_get_size": [_vp, _ip],
    "radtran_f_total_get": [_vp, _ip, _dp],
    "radtran_photons_sol_get_size": [_vp, _ip],
    "radtran_photons_sol_get": [_vp, _ip, _dp],
    "climaradtranwrk_fup_a_get_size": [_vp, _ip, _ip],
    "climaradtranwrk_fup_a_get": [_vp, _ip, _ip, _dp],
    "climaradtranwrk_fdn_a_get_size": [_vp, _ip, _ip],
    "climaradtranwrk_fdn_a_get": [_vp, _ip, _ip, _dp],
    "climaradtranwrk_fup_n_get_size": [_vp, _ip],
    "climaradtranwrk_fup_n_get": [_vp, _ip, _dp],
    "climaradtranwrk_fdn_n_get_size": [_vp, _ip],
    "climaradtranwrk_fdn_n_get": [_vp, _ip, _dp],
    "climaradtranwrk_amean_get_size": [_vp, _ip, _ip],
    "climaradtranwrk_amean_get": [_vp, _ip, _ip, _dp],
    "climaradtranwrk_tau_band_get_size": [_vp, _ip, _ip],
    "climaradtranwrk_tau_band_get": [_vp, _ip, _ip, _dp],
    "rtchannel_wavl_get_size": [_vp, _ip],
    "rtchannel_wavl_get": [_vp, _ip, _dp],
    "rtchannel_freq_get_size": [_vp, _ip],
    "rtchannel_freq_get": [_vp, _ip, _dp],
}

_lib = None


def load():
    """Load libclima_radtran_hip.so; raise loudly when it is absent (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "clima_amd: HIP extension %s is missing. Build it with "
                "`python -m clima_amd.build` (hipcc, gfx950); there is no CPU fallback." % LIB_PATH)
        # PyTorch (device memory, streams, torch.distributed plumbing) bundles its own
        # libamdhip64.so.7; importing it first makes this library bind to the same HIP
        # runtime, so device pointers and the current device are shared with torch.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export it
            fn.argtypes = argtypes
            fn.restype = None
        _lib = L
    return _lib
