"""ctypes binding of libsophtmi.so (the HIP / gfx950 library under csrc/).

There is no CPU fallback: importing this module without the built library raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SOPHT_MPI_AMD_LIB names another build of the same library (used to compare kernel variants)
LIB_PATH = os.environ.get("SOPHT_MPI_AMD_LIB") or os.path.join(_HERE, "lib", "libsophtmi.so")

SMI_F32, SMI_F64 = 0, 1
SMI_POISSON_ROCFFT, SMI_POISSON_PRUNED = 0, 1
SMI_FILTER_MULTIPLICATIVE, SMI_FILTER_CONVOLUTION = 0, 1
SMI_DELTA_COSINE, SMI_DELTA_PESKIN = 0, 1
SMI_SUM, SMI_MIN, SMI_MAX = 0, 1, 2


class SophtMIError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make` at the repository root "
        "(hipcc --offload-arch=gfx950). sopht_mpi_amd has no CPU fallback."
    )

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

_p, _i, _d, _sz, _i64 = C.c_void_p, C.c_int, C.c_double, C.c_size_t, C.c_int64
_pp = C.POINTER(C.c_void_p)
_dp = C.POINTER(C.c_double)
_i64p = C.POINTER(C.c_int64)
EXCHANGE_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int)
ALLTOALL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
SENDRECV_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                          C.c_size_t)
GROUP_ALLTOALL_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int)
P2P_CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_int,
                     C.POINTER(C.c_int), C.POINTER(C.c_void_p), C.c_size_t)

# name -> argtypes; every function returns int status unless listed in _SPECIAL
SIGNATURES = {
    "smi_device_count": [C.POINTER(C.c_int)],
    "smi_ctx_create": [_pp, _i, _i, _i, _i, _i, _i, _i, _i],
    "smi_ctx_destroy": [_p],
    "smi_sync": [_p],
    "smi_ctx_wait_for": [_p, _p],
    "smi_check_forcing_support": [_p, _p, C.POINTER(C.c_longlong)],
    "smi_comm_unique_id": [_p],
    "smi_comm_init": [_p, _p],
    "smi_set_exchange_callback": [_p, EXCHANGE_CB, _p],
    "smi_set_alltoall_callback": [_p, ALLTOALL_CB],
    "smi_ctx_set_pencil": [_p, _i, _i, _i, _i],
    "smi_set_sendrecv_callback": [_p, SENDRECV_CB],
    "smi_set_group_alltoall_callback": [_p, GROUP_ALLTOALL_CB],
    "smi_set_p2p_callback": [_p, P2P_CB],
    "smi_update_vorticity_from_penalised_velocity": [_p, _p, _p, _p, _d],
    "smi_brinkmann_penalise": [_p, _p, _p, _p, _p, _d, _i],
    "smi_char_func_from_level_set": [_p, _p, _p, _d],
    "smi_ctx_set_grid3": [_p, _i, _i, _i, _i, _i, _i],
    "smi_ctx_set_block": [_p, _p, _p, _p],
    "smi_ctx_create_block": [_pp, _i, _i, _i, _i, _i, _i, _i, _i, _p, _p, _p],
    "smi_halo_exchange_general": [_p, _p, _i, _i],
    "smi_poisson_set_pencil_route": [_p, _p, _p, _p, _p, _p],
    "smi_poisson_pencil_solve": [_p, _p, _p, _i],
    "smi_pencil_to_slab": [_p, _p, _p, _p, _i, _p, _p],
    "smi_slab_to_pencil": [_p, _p, _p, _p, _i, _p, _p],
    "smi_allreduce_f64": [_p, _dp, _i, _i],
    "smi_malloc": [_p, _sz, _pp],
    "smi_free": [_p, _p],
    "smi_memcpy_h2d": [_p, _p, _p, _sz],
    "smi_memcpy_d2h": [_p, _p, _p, _sz],
    "smi_memcpy_d2d": [_p, _p, _p, _sz],
    "smi_memset_zero": [_p, _p, _sz],
    "smi_halo_exchange": [_p, _p, _i],
    "smi_planes_d2h": [_p, _p, _i, _i, _i, _p],
    "smi_planes_h2d": [_p, _p, _i, _i, _i, _p, _i],
    "smi_set_fixed_val": [_p, _p, _i, _dp],
    "smi_add_fixed_val": [_p, _p, _p, _i, _dp],
    "smi_elementwise_sum": [_p, _p, _p, _p, _i],
    "smi_elementwise_copy": [_p, _p, _p, _i],
    "smi_elementwise_saxpby": [_p, _p, _p, _d, _p, _d, _i],
    "smi_cross_product": [_p, _p, _p, _p],
    "smi_diffusion_flux": [_p, _p, _p, _d, _i],
    "smi_diffusion_timestep": [_p, _p, _p, _d, _i],
    "smi_update_vorticity_from_velocity_forcing": [_p, _p, _p, _d],
    "smi_curl": [_p, _p, _p, _d],
    "smi_divergence": [_p, _p, _p, _d],
    "smi_penalise_boundary": [_p, _p, _i, _i, _dp, _dp],
    "smi_laplacian_filter": [_p, _p, _i, _p, _p, _i, _i],
    "smi_laplacian_filter_w": [_p, _p, _i, _p, _p, _i, _i, _i],
    "smi_advection_timestep_eno3": [_p, _p, _p, _p, _d, _i],
    "smi_advection_flux_eno3": [_p, _p, _p, _p, _d],
    "smi_rotational_advection_update": [_p, _p, _p, _p, _d],
    "smi_advection_diffusion_update": [_p, _p, _p, _p, _p, _d, _d],
    "smi_reduce_max_abs_sum": [_p, _p, _dp],
    "smi_reduce_max": [_p, _p, _i, _dp],
    "smi_step_velocity_max_abs_sum": [_p, _p, _dp],
    "smi_reduce_sum_sq": [_p, _p, _i, _dp],
    "smi_poisson_create": [_pp, _p, _i, _i, _i, _d, _d, _i],
    "smi_poisson_destroy": [_p],
    "smi_poisson_solve": [_p, _p, _p],
    "smi_poisson_vector_solve": [_p, _p, _p],
    "smi_poisson_solve_noalloc": [_p, _p, _p],
    "smi_poisson_greens_download": [_p, _p],
    "smi_fft_create": [_pp, _p, _i, _i, _i],
    "smi_fft_destroy": [_p],
    "smi_fft_forward": [_p, _p, _p],
    "smi_fft_backward": [_p, _p, _p],
    "smi_copy_to_doubled_domain": [_p, _p, _p],
    "smi_copy_from_doubled_domain": [_p, _p, _p],
    "smi_ib_support": [_p, _p, _p, _p, _i64, _d, _d, _i64p],
    "smi_ib_weights": [_p, _p, _p, _i64, _d, _i],
    "smi_ib_interpolate": [_p, _p, _p, _p, _p, _i64, _i, _d],
    "smi_ib_spread": [_p, _p, _p, _p, _p, _i64, _i],
    "smi_ghost_sum": [_p, _p, _i],
    "smi_clear_ghost_cells": [_p, _p, _i],
    "smi_ib_forcing_fused": [_p, _p, _p, _p, _p, _p, _p, _p, _i64, _d, _d, _i64p, _d, _d, _i],
    "smi_ib_spread_fused": [_p, _p, _p, _p, _i64, _d, _d, _i64p, _i],
    "smi_ib_mismatch_step": [_p, _p, _p, _i64, _d],
    "smi_ib_penalty_force": [_p, _p, _p, _p, _p, _p, _i64, _d, _d],
    "smi_ib_axpby": [_p, _p, _d, _p, _d, _p, _i64],
    "smi_ib_frame_nodes": [_p, _p, _p, _p, _p, _p, _p, _i, _i64],
    "smi_ib_frame_loads": [_p, _p, _p, _p, _p, _i, _i64],
    "smi_set_ordered_spread": [_p, _i],
    "smi_ctx_set_option": [_p, C.c_char_p, _i],
    "smi_ctx_get_option": [_p, C.c_char_p, C.POINTER(C.c_int)],
    "smi_prof_enable": [_p, _i],
    "smi_prof_reset": [_p],
    "smi_prof_get": [_p, _i, C.c_char_p, C.POINTER(C.c_int64), _dp, _dp],
    "smi_prof_get_samples": [_p, _i, _dp, _i, C.POINTER(C.c_int)],
    "smi2d_diffusion_flux": [_p, _p, _p, _d],
    "smi2d_diffusion_timestep": [_p, _p, _p, _d],
    "smi2d_advection_timestep_eno3": [_p, _p, _p, _p, _d],
    "smi2d_advection_flux_eno3": [_p, _p, _p, _p, _d],
    "smi2d_outplane_field_curl": [_p, _p, _p, _d],
    "smi2d_update_vorticity_from_velocity_forcing": [_p, _p, _p, _d],
    "smi2d_penalise_boundary": [_p, _p, _i, _dp, _dp],
    "smi2d_set_fixed_val": [_p, _p, _i, _dp],
    "smi2d_add_fixed_val": [_p, _p, _p, _i, _dp],
    "smi2d_reduce_max_abs_sum": [_p, _p, _dp],
    "smi2d_reduce_max": [_p, _p, _dp],
    "smi2d_poisson_create": [_pp, _p, _p, _d],
    "smi2d_poisson_destroy": [_p],
    "smi2d_poisson_solve": [_p, _p, _p],
    "smi2d_halo_exchange": [_p, _p, _i],
    "smi2d_flow_step_ns": [_p, _p, _p, _p, _p, _p, _p, _d, _d, _d, _i, _dp, _dp, _dp],
    "smi2d_ib_support": [_p, _p, _p, _p, C.c_int64, _d, _d],
    "smi2d_ib_weights": [_p, _p, _p, C.c_int64, _d, _i],
    "smi2d_ib_interpolate": [_p, _p, _p, _p, _p, C.c_int64, _i, _d],
    "smi2d_ib_spread": [_p, _p, _p, _p, _p, C.c_int64, _i, _i],
    "smi2d_ghost_sum": [_p, _p, _i, _i],
    "smi_set_forcing_support": [_p, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "smi_flow_step_ns3d": [_p, _p, _p, _p, _p, _p, _p, _p, _d, _d, _d, _i, _dp, _dp, _dp, _i, _i],
    "smi_flow_step_ns3d_swap": [_p, _p, _p, _p, _p, _p, _p, _p, _d, _d, _d, _i, _dp, _dp, _dp, _i, _i,
                                C.POINTER(C.c_int)],
}
_SPECIAL = {
    "smi_last_error": (C.c_char_p, []),
    "smi_version": (C.c_int, []),
    "smi_prof_num_slots": (C.c_int, []),
    "smi_poisson_greens_size": (C.c_size_t, [_p]),
    "smi_poisson_noalloc_capacity": (C.c_size_t, []),
    "smi_poisson_supports": (C.c_int, [_i, _i, _i, _i, _i]),
}

for _name, _args in SIGNATURES.items():
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = C.c_int
for _name, (_res, _args) in _SPECIAL.items():
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = _res


def last_error():
    return lib.smi_last_error().decode("utf-8", "replace")


def check(status):
    if status != 0:
        raise SophtMIError(last_error())


def call(name, *args):
    """Call an int-status entry point and raise SophtMIError on failure."""
    check(getattr(lib, name)(*args))


def darray(values):
    """Host double array for ABI arguments typed `const double*`."""
    vals = [float(v) for v in values]
    return (C.c_double * len(vals))(*vals)


def i64array(values):
    vals = [int(v) for v in values]
    return (C.c_int64 * len(vals))(*vals)


def prof_collect(ctx):
    """Per-slot event timings of the in-library profiler: name -> dict(launches, ms, bytes)."""
    out = {}
    for slot in range(lib.smi_prof_num_slots()):
        name = C.create_string_buffer(64)
        n, ms, by = C.c_int64(0), C.c_double(0.0), C.c_double(0.0)
        call("smi_prof_get", ctx.handle, slot, name, C.byref(n), C.byref(ms), C.byref(by))
        if n.value:
            cap = min(int(n.value), 4096)
            buf, cnt = (C.c_double * cap)(), C.c_int(0)
            call("smi_prof_get_samples", ctx.handle, slot, buf, cap, C.byref(cnt))
            out[name.value.decode()] = {"launches": n.value, "ms": ms.value, "alg_bytes": by.value,
                                        "samples_ms": list(buf[: cnt.value])}
    return out
