"""ctypes binding of libsoicp.so (include/so_icp.h).

This is the thin host-side mirror used by tests and bench.py.  The reference's host is C++
(LidarSLAM, src/LidarProcess/LidarSlam.cpp); its C++ adapter is shown in INTEGRATION.md.  Loading
fails loudly when the library is missing or when no HIP device is usable -- there is no CPU path."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libsoicp.so")
N_REJECT, N_OBS, MAX_OUTER, UID_BYTES, PEER_HANDLE_BYTES = 7, 9, 16, 128, 80

OK, NOT_ENOUGH_MAP_FEATURES, MAP_SEEDED = 0, 1, 2
SHARD_MAP, SHARD_QUERIES = 0, 1


class Config(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device_id", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32),
                ("max_iterations", C.c_int32), ("lm_max_iterations", C.c_int32), ("max_surface_features", C.c_int32),
                ("k", C.c_int32), ("tukey_variant", C.c_int32), ("time_kernels", C.c_int32),
                ("line_res", C.c_float), ("plane_res", C.c_float), ("yaw_ratio", C.c_double),
                ("velocity_failure_threshold", C.c_double), ("shard_mode", C.c_int32), ("solve_workgroups", C.c_int32)]


class IterStats(C.Structure):
    _fields_ = [("translation_norm", C.c_double), ("rotation_norm", C.c_double), ("num_surf_from_scan", C.c_int32),
                ("num_corner_from_scan", C.c_int32), ("lm_iterations", C.c_int32), ("num_successful_steps", C.c_int32),
                ("termination", C.c_int32), ("reserved", C.c_int32), ("initial_cost", C.c_double), ("final_cost", C.c_double),
                ("reject_hist", C.c_int32 * N_REJECT), ("obs_hist", C.c_int32 * N_OBS), ("pose_after", C.c_double * 7)]


class Stats(C.Structure):
    _fields_ = [("laser_cloud_surf_from_map_num", C.c_int32), ("laser_cloud_corner_from_map_num", C.c_int32),
                ("laser_cloud_surf_stack_num", C.c_int32), ("laser_cloud_corner_stack_num", C.c_int32),
                ("n_iterations", C.c_int32), ("startup_count", C.c_int32), ("pos_in_localmap", C.c_int32 * 3),
                ("prediction_source", C.c_int32), ("total_translation", C.c_double), ("total_rotation", C.c_double),
                ("translation_from_last", C.c_double), ("rotation_from_last", C.c_double), ("time_elapsed_ms", C.c_double),
                ("uncertainty", C.c_double * 6), ("JtJ", C.c_double * 36), ("Jtr", C.c_double * 6),
                ("iterations", IterStats * MAX_OUTER), ("flags", C.c_uint32), ("reserved", C.c_uint32)]


# so_icp_stats.flags
FLAG_PER_EVAL_LAUNCHES, FLAG_RETRIED, FLAG_HOST_MAP, FLAG_SORT_BINNING, FLAG_SHARDED, FLAG_STAGED_SCAN, FLAG_COPY_READBACK = 1, 2, 4, 8, 16, 32, 64
FLAG_QUERY_SPLIT = 128
FLAG_BINNED_AHEAD = 256
FLAG_QUERY_WAVES = 512
FLAG_CHAINED = 1024


class RegistrationError(C.Structure):
    """so_icp_registration_error_t (LidarSLAM::RegistrationError, LS.h:127-151)"""
    _fields_ = [("covariance", C.c_double * 36), ("position_error", C.c_double), ("position_error_direction", C.c_double * 3),
                ("pos_inverse_condition_num", C.c_double), ("orientation_error_deg", C.c_double),
                ("orientation_error_direction", C.c_double * 3), ("ori_inverse_condition_num", C.c_double)]


class PrefilterInfo(C.Structure):
    _fields_ = [("average_distance", C.c_double), ("count_far_points", C.c_int32), ("increase_blind_radius", C.c_int32),
                ("line_res", C.c_float), ("plane_res", C.c_float), ("statistic_in_input_order", C.c_int32), ("reserved", C.c_int32)]


class DeskewInfo(C.Structure):
    _fields_ = [("q_w_original_l", C.c_double * 4), ("t_w_original_l", C.c_double * 3), ("n_clamped", C.c_uint32), ("reserved", C.c_uint32)]


class Timing(C.Structure):
    _fields_ = [("knn_ms_total", C.c_double), ("knn_launches", C.c_int64), ("knn_queries", C.c_int64), ("knn_map_points", C.c_int64),
                ("eval_ms_total", C.c_double), ("eval_launches", C.c_int64), ("eval_points", C.c_int64),
                ("prep_ms_total", C.c_double), ("prep_launches", C.c_int64),
                ("host_ms_total", C.c_double), ("registrations", C.c_int64),
                ("knn_group_passes", C.c_int64), ("knn_fallback_lanes", C.c_int64), ("knn_candidates_scanned", C.c_int64),
                ("stage_wait_ms_total", C.c_double), ("staged_direct", C.c_int64), ("staged_copied", C.c_int64), ("stage_declined", C.c_int64),
                ("knn_packed_rows", C.c_int64), ("knn_packed_rows_too_many_runs", C.c_int64), ("knn_packed_rows_tile_full", C.c_int64),
                ("knn_packed_kept", C.c_int64), ("knn_pack_registrations", C.c_int64), ("knn_pack_holds", C.c_int64),
                ("seq_chained", C.c_int64), ("seq_chain_breaks", C.c_int64)]


class Sums(C.Structure):
    _fields_ = [("cost", C.c_double), ("count", C.c_double), ("Jtr", C.c_double * 6), ("JtJ", C.c_double * 21), ("hist", C.c_double * 16)]


class LmState(C.Structure):
    _fields_ = [("opaque", C.c_double * 96)]


EXPORTED = ["so_icp_default_config", "so_icp_create", "so_icp_destroy", "so_icp_last_error", "so_icp_abi_version",
            "so_icp_device_available", "so_icp_set_resolution", "so_icp_set_max_surface_features", "so_icp_set_max_iterations",
            "so_icp_map_set_origin", "so_icp_map_shift", "so_icp_map_add_surf", "so_icp_map_count_5x5", "so_icp_map_export",
            "so_icp_map_size", "so_icp_map_clear", "so_icp_map_get_origin", "so_icp_knn_surf", "so_icp_register",
            "so_icp_register_dev", "so_icp_upload_scan", "so_icp_free_scan", "so_icp_localization", "so_icp_comm_unique_id", "so_icp_comm_init",
            "so_icp_shard_owner_of_point", "so_icp_cells_per_cube", "so_icp_lm_begin", "so_icp_lm_feed", "so_icp_lm_result",
            "so_icp_get_timing", "so_icp_reset_timing", "so_icp_set_time_kernels", "so_icp_synchronize", "so_icp_debug_stamps", "so_icp_debug_knn_stamps", "so_icp_register_batch", "so_icp_registration_error",
            "so_icp_localization_dev", "so_icp_download_scan", "so_icp_prefilter_scan", "so_icp_stage_scan", "so_icp_debug_match_status", "so_icp_comm_init_inprocess", "so_icp_peer_export", "so_icp_peer_connect", "so_icp_peer_enable",
            "so_icp_deskew_scan", "so_icp_deskew_scan_dev", "so_icp_transform_cloud", "so_icp_shard_histogram",
            "so_icp_host_register", "so_icp_host_unregister", "so_icp_host_alloc", "so_icp_host_free", "so_icp_device_count", "so_icp_stage_cancel",
            "so_icp_map_insert_stats", "so_icp_register_sequence", "so_icp_map_export_records", "so_icp_sequence_announce_next", "so_icp_debug_neighbours", "so_icp_prefilter_announce"]

_lib = None


def load():
    """Load libsoicp.so; raise if it is absent (run `python -m superodom_amd.build`)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} not built: run `python -m superodom_amd.build` (hipcc, gfx950). "
                           "superodom_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    vp, f32p, f64p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int32)
    u8p = C.POINTER(C.c_uint8)
    L.so_icp_default_config.argtypes = [C.POINTER(Config)]; L.so_icp_default_config.restype = None
    L.so_icp_create.argtypes = [C.POINTER(Config)]; L.so_icp_create.restype = vp
    L.so_icp_destroy.argtypes = [vp]; L.so_icp_destroy.restype = None
    L.so_icp_last_error.argtypes = [vp]; L.so_icp_last_error.restype = C.c_char_p
    L.so_icp_set_resolution.argtypes = [vp, C.c_float, C.c_float]
    L.so_icp_set_max_surface_features.argtypes = [vp, C.c_int]
    L.so_icp_set_max_iterations.argtypes = [vp, C.c_int]
    L.so_icp_map_set_origin.argtypes = [vp, f64p, i32p]
    L.so_icp_map_shift.argtypes = [vp, f64p, i32p]
    L.so_icp_map_add_surf.argtypes = [vp, f32p, C.c_size_t, C.c_size_t]
    L.so_icp_map_count_5x5.argtypes = [vp, i32p, i32p, i32p]
    L.so_icp_map_export.argtypes = [vp, f32p, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, i32p]
    L.so_icp_map_size.argtypes = [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.so_icp_map_export_records.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_size_t), C.c_int, i32p]
    L.so_icp_map_clear.argtypes = [vp]
    L.so_icp_map_insert_stats.argtypes = [vp, C.POINTER(C.c_uint), C.POINTER(C.c_uint)]
    L.so_icp_map_get_origin.argtypes = [vp, i32p]
    L.so_icp_knn_surf.argtypes = [vp, f32p, C.c_size_t, C.c_int, f32p, f32p, i32p, u8p]
    L.so_icp_register.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, f64p, f64p, C.POINTER(Stats)]
    L.so_icp_register_dev.argtypes = [vp, vp, C.c_size_t, f64p, f64p, C.POINTER(Stats)]
    L.so_icp_sequence_announce_next.argtypes = [vp, f32p, C.c_size_t, f64p]
    L.so_icp_register_sequence.argtypes = [vp, C.c_int, C.POINTER(vp), C.POINTER(C.c_size_t), C.c_size_t, C.c_int, f64p, f64p, f64p, f64p, C.POINTER(Stats), i32p]
    L.so_icp_upload_scan.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.POINTER(vp)]
    L.so_icp_free_scan.argtypes = [vp, vp]
    L.so_icp_localization.argtypes = [vp, C.c_int, f64p, f32p, C.c_size_t, C.c_size_t, C.c_double, f64p, C.POINTER(Stats)]
    L.so_icp_comm_unique_id.argtypes = [u8p]
    L.so_icp_comm_init.argtypes = [vp, u8p]
    L.so_icp_shard_owner_of_point.argtypes = [f32p, i32p, C.c_float, C.c_int]
    L.so_icp_cells_per_cube.argtypes = [C.c_float, f64p]
    L.so_icp_shard_histogram.argtypes = [f32p, C.c_size_t, C.c_size_t, f64p, i32p, C.c_float, C.c_int, C.POINTER(C.c_int64)]
    L.so_icp_lm_begin.argtypes = [C.POINTER(LmState), f64p, C.POINTER(Sums), C.c_int, f64p]
    L.so_icp_lm_feed.argtypes = [C.POINTER(LmState), C.POINTER(Sums), f64p]
    L.so_icp_lm_result.argtypes = [C.POINTER(LmState), f64p, C.POINTER(IterStats)]
    L.so_icp_get_timing.argtypes = [vp, C.POINTER(Timing)]
    L.so_icp_reset_timing.argtypes = [vp]
    L.so_icp_synchronize.argtypes = [vp]
    L.so_icp_debug_stamps.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.so_icp_register_batch.argtypes = [vp, f32p, vp, C.c_size_t, C.c_size_t, f64p, C.c_int, f64p, C.POINTER(Stats), i32p]
    L.so_icp_registration_error.argtypes = [C.POINTER(Stats), C.POINTER(RegistrationError)]
    L.so_icp_localization_dev.argtypes = [vp, C.c_int, f64p, vp, C.c_size_t, C.c_double, f64p, C.POINTER(Stats)]
    L.so_icp_download_scan.argtypes = [vp, vp, C.c_size_t, f32p]
    for fn in (L.so_icp_deskew_scan, L.so_icp_deskew_scan_dev):
        fn.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_double, C.POINTER(C.c_double), C.c_size_t, C.c_int, C.POINTER(C.c_double),
                       C.POINTER(DeskewInfo)]
    L.so_icp_transform_cloud.argtypes = [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_uint8), C.POINTER(C.c_size_t)]
    L.so_icp_prefilter_announce.argtypes = [vp, f32p, C.c_size_t, C.c_size_t]
    L.so_icp_prefilter_scan.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.c_int, C.c_float, C.c_float, C.POINTER(vp),
                                        C.POINTER(C.c_size_t), C.POINTER(PrefilterInfo)]
    L.so_icp_debug_knn_stamps.argtypes = [vp, C.POINTER(C.c_uint64), C.c_size_t, C.POINTER(C.c_size_t)]
    L.so_icp_set_time_kernels.argtypes = [vp, C.c_int]
    L.so_icp_stage_scan.argtypes = [vp, f32p, C.c_size_t, C.c_size_t]
    L.so_icp_host_register.argtypes = [vp, vp, C.c_size_t]
    L.so_icp_stage_cancel.argtypes = [vp, f32p]
    L.so_icp_host_unregister.argtypes = [vp, vp]
    L.so_icp_host_alloc.argtypes = [vp, C.c_size_t, C.POINTER(vp)]
    L.so_icp_host_free.argtypes = [vp, vp]
    L.so_icp_debug_match_status.argtypes = [vp, u8p, C.c_size_t]
    L.so_icp_debug_neighbours.argtypes = [vp, C.POINTER(C.c_uint32), C.c_size_t]
    L.so_icp_comm_init_inprocess.argtypes = [vp, C.c_uint64]
    L.so_icp_peer_export.argtypes = [vp, u8p]
    L.so_icp_peer_connect.argtypes = [vp, u8p, i32p]
    L.so_icp_peer_enable.argtypes = [vp, C.c_int]
    _lib = L
    return L


_F64P = C.POINTER(C.c_double)


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def default_config(**kw):
    cfg = Config()
    load().so_icp_default_config(C.byref(cfg))
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


class SoIcpError(RuntimeError):
    pass


class LidarSlamGpu:
    """Host-side mirror of the reference's LidarSLAM object for this path: owns one so_icp_ctx
    (= LidarSLAM::localMap + the ICP state), methods named after the reference members they replace."""

    def __init__(self, **cfg_kw):
        self.L = load()
        self.cfg = default_config(**cfg_kw)
        self.h = self.L.so_icp_create(C.byref(self.cfg))
        if not self.h:
            raise SoIcpError(self.L.so_icp_last_error(None).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.so_icp_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc < 0:
            raise SoIcpError(f"so_icp error {rc}: {self.L.so_icp_last_error(self.h).decode()}")
        return rc

    # ---- LocalMap ----
    def set_resolution(self, line_res, plane_res):
        self._check(self.L.so_icp_set_resolution(self.h, line_res, plane_res))

    def set_max_surface_features(self, n):
        self._check(self.L.so_icp_set_max_surface_features(self.h, int(n)))

    def set_max_iterations(self, n):
        self._check(self.L.so_icp_set_max_iterations(self.h, int(n)))

    def set_origin(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64); o = np.zeros(3, np.int32)
        self._check(self.L.so_icp_map_set_origin(self.h, _p(t, C.c_double), _p(o, C.c_int32)))
        return o

    def origin(self):
        o = np.zeros(3, np.int32); self._check(self.L.so_icp_map_get_origin(self.h, _p(o, C.c_int32))); return o

    def shift_map(self, t):
        t = np.ascontiguousarray(t, dtype=np.float64); o = np.zeros(3, np.int32)
        self._check(self.L.so_icp_map_shift(self.h, _p(t, C.c_double), _p(o, C.c_int32)))
        return o

    def add_surf_point_cloud(self, xyz):
        xyz = _f32(xyz).reshape(-1, 3)
        return self._check(self.L.so_icp_map_add_surf(self.h, _p(xyz, C.c_float), len(xyz), 12))

    def count_5x5(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.int32); ne = C.c_int32(); ns = C.c_int32()
        self._check(self.L.so_icp_map_count_5x5(self.h, _p(pos, C.c_int32), C.byref(ne), C.byref(ns)))
        return ns.value

    def map_size(self, this_rank=False):
        n = C.c_size_t(); nr = C.c_size_t()
        self._check(self.L.so_icp_map_size(self.h, C.byref(n), C.byref(nr) if this_rank else None))
        return (n.value, nr.value) if this_rank else n.value

    def export_map(self, only_5x5=False, pos=None):
        n = self.map_size(); out = np.zeros((max(n, 1), 3), np.float32); m = C.c_size_t()
        posa = None if pos is None else np.ascontiguousarray(pos, dtype=np.int32)
        self._check(self.L.so_icp_map_export(self.h, _p(out, C.c_float), n, C.byref(m), int(only_5x5),
                                             None if posa is None else _p(posa, C.c_int32)))
        return out[:m.value].copy()

    def export_map_records(self, stride=32, only_5x5=False, pos=None, out=None):
        """so_icp_map_export_records: the map as records of `stride` bytes (x, y, z floats first, rest zero) -> uint8 array (n, stride)"""
        posa = None if pos is None else np.ascontiguousarray(pos, dtype=np.int32)
        pp = None if posa is None else _p(posa, C.c_int32)
        m = C.c_size_t()
        self._check(self.L.so_icp_map_export_records(self.h, None, stride, 0, C.byref(m), int(only_5x5), pp))
        n = m.value
        buf = out if out is not None else np.zeros((max(n, 1), stride), np.uint8)
        assert buf.nbytes >= n * stride
        self._check(self.L.so_icp_map_export_records(self.h, buf.ctypes.data_as(C.c_void_p), stride, n, C.byref(m), int(only_5x5), pp))
        return buf.reshape(-1)[:m.value * stride].reshape(m.value, stride)

    def clear_map(self):
        self._check(self.L.so_icp_map_clear(self.h))

    def map_insert_stats(self):
        """(inserts laid out by the device, of those handed back to the host's round-by-round path)"""
        a = C.c_uint(); b = C.c_uint()
        self._check(self.L.so_icp_map_insert_stats(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    # ---- Seam B ----
    def nearest_k_search_surf(self, q, k=5, want_index=True):
        q = _f32(q).reshape(-1, 3); nq = len(q)
        nbr = np.zeros((nq, k, 3), np.float32); d2 = np.zeros((nq, k), np.float32)
        idx = np.zeros((nq, k), np.int32); found = np.zeros(nq, np.uint8)
        self._check(self.L.so_icp_knn_surf(self.h, _p(q, C.c_float), nq, k, _p(nbr, C.c_float), _p(d2, C.c_float),
                                           _p(idx, C.c_int32) if want_index else None, _p(found, C.c_uint8)))
        return found, nbr, d2, idx

    # ---- Seam A ----
    def register(self, scan, pose_in):
        scan = _f32(scan).reshape(-1, 3); pose_in = np.ascontiguousarray(pose_in, dtype=np.float64)
        out = np.zeros(7); st = Stats()
        rc = self._check(self.L.so_icp_register(self.h, _p(scan, C.c_float), len(scan), 12, _p(pose_in, C.c_double),
                                                _p(out, C.c_double), C.byref(st)))
        return rc, out, st

    def stage_scan(self, scan):
        """so_icp_stage_scan: announce the NEXT scan (contiguous float32 (n,3) array that stays alive and unchanged until
        the register call that consumes it)."""
        assert isinstance(scan, np.ndarray) and scan.dtype == np.float32 and scan.flags.c_contiguous
        self._check(self.L.so_icp_stage_scan(self.h, _p(scan, C.c_float), len(scan), 12))

    def stage_cancel(self, scan):
        self._check(self.L.so_icp_stage_cancel(self.h, _p(scan, C.c_float)))

    def host_register(self, arr):
        """so_icp_host_register: pin the numpy array's memory (it must stay alive until host_unregister / close); packed scans
        announced from inside it travel to HBM by DMA straight from the array."""
        assert isinstance(arr, np.ndarray) and arr.flags.c_contiguous
        self._check(self.L.so_icp_host_register(self.h, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def host_alloc_like(self, arr):
        """A copy of `arr` in pinned host memory from so_icp_host_alloc (numpy view; valid until close())."""
        arr = np.ascontiguousarray(arr)
        p = C.c_void_p()
        self._check(self.L.so_icp_host_alloc(self.h, arr.nbytes, C.byref(p)))
        buf = (C.c_char * arr.nbytes).from_address(p.value)
        out = np.frombuffer(buf, dtype=arr.dtype).reshape(arr.shape)
        out[...] = arr
        return out

    def host_unregister(self, arr):
        self._check(self.L.so_icp_host_unregister(self.h, arr.ctypes.data_as(C.c_void_p)))

    def prepare_stage_scan(self, scan):
        assert isinstance(scan, np.ndarray) and scan.dtype == np.float32 and scan.flags.c_contiguous
        fn, args = self.L.so_icp_stage_scan, (self.h, _p(scan, C.c_float), len(scan), 12)
        return lambda: fn(*args)

    def prepare_register(self, scan, pose_in, stats, pose_out):
        """Zero-argument callable performing one so_icp_register (host scan buffer) with pre-built ctypes arguments."""
        assert isinstance(scan, np.ndarray) and scan.dtype == np.float32 and scan.flags.c_contiguous
        assert pose_in.dtype == np.float64 and pose_in.flags.c_contiguous and pose_out.dtype == np.float64 and pose_out.flags.c_contiguous
        fn, args = self.L.so_icp_register, (self.h, _p(scan, C.c_float), len(scan), 12, pose_in.ctypes.data_as(_F64P),
                                            pose_out.ctypes.data_as(_F64P), C.byref(stats))
        return lambda: fn(*args)

    def match_status(self, n):
        """MatchingResult of every query of the last registration's last outer iteration (so_icp_debug_match_status)."""
        out = np.zeros(n, np.uint8)
        self._check(self.L.so_icp_debug_match_status(self.h, _p(out, C.c_uint8), n))
        return out

    def neighbours(self, n):
        """the five neighbours (canonical map indices) the last k-NN sweep left for every query (so_icp_debug_neighbours)"""
        out = np.zeros((n, 5), np.uint32)
        self._check(self.L.so_icp_debug_neighbours(self.h, _p(out, C.c_uint32), n))
        return out

    def upload_scan(self, scan):
        scan = _f32(scan).reshape(-1, 3); d = C.c_void_p()
        self._check(self.L.so_icp_upload_scan(self.h, _p(scan, C.c_float), len(scan), 12, C.byref(d)))
        return d.value, len(scan)

    def free_scan(self, d_scan):
        self._check(self.L.so_icp_free_scan(self.h, d_scan))

    def register_dev(self, d_scan, n, pose_in, stats=None, pose_out=None):
        """pose_in / pose_out may be preallocated contiguous float64 arrays (no per-call allocation in a timed loop)."""
        if not (isinstance(pose_in, np.ndarray) and pose_in.dtype == np.float64 and pose_in.flags.c_contiguous):
            pose_in = np.ascontiguousarray(pose_in, dtype=np.float64)
        out = pose_out if pose_out is not None else np.zeros(7)
        st = stats if stats is not None else Stats()
        rc = self.L.so_icp_register_dev(self.h, d_scan, n, pose_in.ctypes.data_as(_F64P), out.ctypes.data_as(_F64P), C.byref(st))
        if rc < 0:
            self._check(rc)
        return rc, out, st

    def prepare_register_dev(self, d_scan, n, pose_in, stats, pose_out):
        """A zero-argument callable that performs exactly one so_icp_register_dev call with pre-built ctypes arguments
        (timed loops: no per-call argument conversion in Python).  pose_in / pose_out: contiguous float64 arrays that
        stay alive; returns the C return code."""
        assert pose_in.dtype == np.float64 and pose_in.flags.c_contiguous and pose_out.dtype == np.float64 and pose_out.flags.c_contiguous
        fn, args = self.L.so_icp_register_dev, (self.h, d_scan, n, pose_in.ctypes.data_as(_F64P), pose_out.ctypes.data_as(_F64P), C.byref(stats))
        return lambda: fn(*args)

    def prepare_register_sequence(self, scans, pose0, deltas, on_device=False):
        """so_icp_register_sequence with pre-built arguments.  scans: contiguous float32 (n, 3) host arrays, or (device pointer, n) pairs
        with on_device.  deltas: (count, 7), row 0 unused.  Returns (call, poses_out (count, 7), guesses_out (count, 7), stats array, n_done):
        call() performs the C call and returns its code."""
        count = len(scans)
        ptrs = (C.c_void_p * count)(); ns = (C.c_size_t * count)()
        for k, sc in enumerate(scans):
            if on_device:
                ptrs[k], ns[k] = sc[0], sc[1]
            else:
                assert isinstance(sc, np.ndarray) and sc.dtype == np.float32 and sc.flags.c_contiguous
                ptrs[k], ns[k] = sc.ctypes.data, len(sc)
        pose0 = np.ascontiguousarray(pose0, dtype=np.float64); deltas = np.ascontiguousarray(deltas, dtype=np.float64).reshape(count, 7)
        out = np.zeros((count, 7)); guesses = np.zeros((count, 7)); st = (Stats * count)(); n_done = C.c_int32(0)
        fn, args = self.L.so_icp_register_sequence, (self.h, count, ptrs, ns, 12, 1 if on_device else 0, pose0.ctypes.data_as(_F64P),
                                                     deltas.ctypes.data_as(_F64P), out.ctypes.data_as(_F64P), guesses.ctypes.data_as(_F64P), st, C.byref(n_done))
        keep = (ptrs, ns, pose0, deltas, scans)
        return (lambda: fn(*args)), out, guesses, st, n_done, keep

    def sequence_announce_next(self, scan, delta):
        """so_icp_sequence_announce_next: the scan that will start the NEXT register_sequence call (None withdraws)"""
        if scan is None:
            self._check(self.L.so_icp_sequence_announce_next(self.h, None, 0, None)); return
        assert isinstance(scan, np.ndarray) and scan.dtype == np.float32 and scan.flags.c_contiguous
        d = np.ascontiguousarray(delta, dtype=np.float64)
        self._check(self.L.so_icp_sequence_announce_next(self.h, _p(scan, C.c_float), len(scan), _p(d, C.c_double)))

    def register_sequence(self, scans, pose0, deltas, on_device=False):
        call, out, guesses, st, n_done, _keep = self.prepare_register_sequence(scans, pose0, deltas, on_device)
        rc = call()
        if rc < 0:
            self._check(rc)
        return rc, out, guesses, list(st), n_done.value

    def register_batch(self, scan, poses_in, d_scan=None, n=None):
        """Same scan, many initial poses (so_icp_register_batch).  scan: host array, or None with (d_scan, n) from upload_scan.
        Returns (number of hypotheses that converged normally, rc[B], poses_out[B,7], stats list)."""
        poses_in = np.ascontiguousarray(poses_in, dtype=np.float64).reshape(-1, 7)
        B = len(poses_in)
        out = np.zeros((B, 7)); rcs = np.zeros(B, np.int32); st = (Stats * B)()
        if d_scan is None:
            scan = _f32(scan).reshape(-1, 3)
            ok = self._check(self.L.so_icp_register_batch(self.h, _p(scan, C.c_float), None, len(scan), 12, _p(poses_in, C.c_double), B,
                                                          _p(out, C.c_double), st, _p(rcs, C.c_int32)))
        else:
            ok = self._check(self.L.so_icp_register_batch(self.h, None, d_scan, n, 12, _p(poses_in, C.c_double), B,
                                                          _p(out, C.c_double), st, _p(rcs, C.c_int32)))
        return ok, rcs, out, list(st)

    def localization(self, initialization, T_w_lidar, planar_points, time_laser_odometry):
        scan = _f32(planar_points).reshape(-1, 3); T = np.ascontiguousarray(T_w_lidar, dtype=np.float64)
        out = np.zeros(7); st = Stats()
        rc = self._check(self.L.so_icp_localization(self.h, int(bool(initialization)), _p(T, C.c_double), _p(scan, C.c_float),
                                                    len(scan), 12, float(time_laser_odometry), _p(out, C.c_double), C.byref(st)))
        return rc, out, st

    def localization_dev(self, initialization, T_w_lidar, d_scan, n, time_laser_odometry):
        T = np.ascontiguousarray(T_w_lidar, dtype=np.float64); out = np.zeros(7); st = Stats()
        rc = self._check(self.L.so_icp_localization_dev(self.h, int(bool(initialization)), _p(T, C.c_double), d_scan, n,
                                                        float(time_laser_odometry), _p(out, C.c_double), C.byref(st)))
        return rc, out, st

    def download_scan(self, d_scan, n):
        out = np.zeros((n, 3), np.float32)
        self._check(self.L.so_icp_download_scan(self.h, d_scan, n, _p(out, C.c_float)))
        return out

    def prefilter_announce(self, surf_points):
        """so_icp_prefilter_announce: the raw cloud of the NEXT prefilter_scan call starts its way to HBM now (None withdraws)"""
        if surf_points is None:
            self._check(self.L.so_icp_prefilter_announce(self.h, None, 0, 12)); return
        assert isinstance(surf_points, np.ndarray) and surf_points.dtype == np.float32 and surf_points.flags.c_contiguous
        self._check(self.L.so_icp_prefilter_announce(self.h, _p(surf_points, C.c_float), len(surf_points.reshape(-1, 3)), 12))

    def prefilter_scan(self, surf_points, auto_voxel_size, line_res, plane_res):
        """laserMapping::adjustVoxelSize on the device; returns (d_scan, n, PrefilterInfo)."""
        pts = _f32(surf_points).reshape(-1, 3); d = C.c_void_p(); n = C.c_size_t(0); info = PrefilterInfo()
        self._check(self.L.so_icp_prefilter_scan(self.h, _p(pts, C.c_float), len(pts), 12, int(bool(auto_voxel_size)), float(line_res),
                                                 float(plane_res), C.byref(d), C.byref(n), C.byref(info)))
        return d.value, n.value, info

    def deskew_scan(self, records, time_off, lidar_start_time, poses, poses_are_imu, T_i_l=None):
        """featureExtraction::removePointDistortion on the device.  records: uint8 [n, stride] (float x y z at 0 4 8, float time
        at time_off); poses: [n_poses, 8] = time, position, quaternion x y z w.  Returns (rewritten records, DeskewInfo)."""
        rec = np.ascontiguousarray(records, np.uint8).copy()
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 8)
        til = None if T_i_l is None else np.ascontiguousarray(T_i_l, np.float64)
        info = DeskewInfo()
        self._check(self.L.so_icp_deskew_scan(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0], rec.shape[1], int(time_off),
                                              float(lidar_start_time), _p(poses, C.c_double), len(poses), int(bool(poses_are_imu)),
                                              None if til is None else _p(til, C.c_double), C.byref(info)))
        return rec, info

    def transform_cloud(self, records, T_w_lidar):
        """the node's registered scan (pointAssociateToMap over a cloud): records uint8 [n, stride], float x y z at 0 4 8.
        Returns (rewritten records, keep flags [n] uint8, number kept)."""
        rec = np.ascontiguousarray(records, np.uint8).copy()
        T = np.ascontiguousarray(T_w_lidar, np.float64)
        keep = np.zeros(len(rec), np.uint8); nk = C.c_size_t(0)
        self._check(self.L.so_icp_transform_cloud(self.h, rec.ctypes.data_as(C.c_void_p), rec.shape[0], rec.shape[1], _p(T, C.c_double),
                                                  _p(keep, C.c_uint8), C.byref(nk)))
        return rec, keep, nk.value

    def deskew_scan_dev(self, d_records, n, stride, time_off, lidar_start_time, poses, poses_are_imu, T_i_l=None):
        """the same on records resident in HBM (d_records: device address), rewritten there; returns DeskewInfo"""
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 8)
        til = None if T_i_l is None else np.ascontiguousarray(T_i_l, np.float64)
        info = DeskewInfo()
        self._check(self.L.so_icp_deskew_scan_dev(self.h, C.c_void_p(d_records), int(n), int(stride), int(time_off), float(lidar_start_time),
                                                  _p(poses, C.c_double), len(poses), int(bool(poses_are_imu)),
                                                  None if til is None else _p(til, C.c_double), C.byref(info)))
        return info

    # ---- multi-GPU ----
    def comm_init(self, uid_bytes):
        uid = np.frombuffer(bytes(uid_bytes), dtype=np.uint8).copy()
        self._check(self.L.so_icp_comm_init(self.h, _p(uid, C.c_uint8)))

    def comm_init_inprocess(self, group_key):
        self._check(self.L.so_icp_comm_init_inprocess(self.h, int(group_key)))

    def peer_export(self):
        h = np.zeros(PEER_HANDLE_BYTES, np.uint8)
        self._check(self.L.so_icp_peer_export(self.h, _p(h, C.c_uint8)))
        return h.tobytes()

    def peer_connect(self, handles):
        """handles: list of world_size byte strings in rank order; returns the self-test result (bool)."""
        buf = np.frombuffer(b"".join(bytes(h) for h in handles), dtype=np.uint8).copy()
        ok = C.c_int32(0)
        self._check(self.L.so_icp_peer_connect(self.h, _p(buf, C.c_uint8), C.byref(ok)))
        return bool(ok.value)

    def peer_enable(self, on):
        self._check(self.L.so_icp_peer_enable(self.h, int(bool(on))))

    # ---- measurement ----
    def timing(self):
        t = Timing(); self._check(self.L.so_icp_get_timing(self.h, C.byref(t))); return t

    def last_error(self):
        """Text of the context's last error or notice (so_icp_last_error)."""
        return self.L.so_icp_last_error(self.h).decode()

    def reset_timing(self):
        self._check(self.L.so_icp_reset_timing(self.h))

    def set_time_kernels(self, mode):
        self._check(self.L.so_icp_set_time_kernels(self.h, int(mode)))

    def debug_stamps(self):
        out = np.zeros(16, np.uint64)
        self._check(self.L.so_icp_debug_stamps(self.h, _p(out, C.c_uint64)))
        return out

    def debug_knn_stamps(self):
        """(2, workgroups, 16) uint64 records of the k-NN kernel phases (SOICP_ABLATE=128), or None."""
        n = C.c_size_t(0)
        self._check(self.L.so_icp_debug_knn_stamps(self.h, None, 0, C.byref(n)))
        if n.value == 0:
            return None
        out = np.zeros(n.value, np.uint64)
        self._check(self.L.so_icp_debug_knn_stamps(self.h, _p(out, C.c_uint64), n.value, C.byref(n)))
        return out.reshape(2, -1, 16)

    def synchronize(self):
        self._check(self.L.so_icp_synchronize(self.h))


def registration_error(stats):
    """EstimateRegistrationError (LS.cpp:854-889) from the final normal equations; None when J^T J is singular."""
    out = RegistrationError()
    rc = load().so_icp_registration_error(C.byref(stats), C.byref(out))
    return out if rc == 0 else None


def device_count():
    return int(load().so_icp_device_count())


def comm_unique_id():
    uid = np.zeros(UID_BYTES, np.uint8)
    rc = load().so_icp_comm_unique_id(_p(uid, C.c_uint8))
    if rc:
        raise SoIcpError(f"so_icp_comm_unique_id: {load().so_icp_last_error(None).decode()}")
    return uid.tobytes()


def cells_per_cube(plane_res):
    cell = C.c_double()
    nc = load().so_icp_cells_per_cube(float(plane_res), C.byref(cell))
    return nc, cell.value


def shard_histogram(scan, pose, origin, plane_res, world_size):
    """Queries of `scan` (sensor frame) that fall to each rank under `pose` (so_icp_shard_histogram)."""
    scan = _f32(scan).reshape(-1, 3); pose = np.ascontiguousarray(pose, dtype=np.float64); o = np.ascontiguousarray(origin, dtype=np.int32)
    out = np.zeros(int(world_size), np.int64)
    rc = load().so_icp_shard_histogram(_p(scan, C.c_float), len(scan), 12, _p(pose, C.c_double), _p(o, C.c_int32), float(plane_res), int(world_size),
                                       _p(out, C.c_int64))
    if rc:
        raise SoIcpError(f"so_icp_shard_histogram: {rc}")
    return out


def shard_owner_of_point(p, origin, plane_res, world_size):
    p = _f32(p); o = np.ascontiguousarray(origin, dtype=np.int32)
    return load().so_icp_shard_owner_of_point(_p(p, C.c_float), _p(o, C.c_int32), float(plane_res), int(world_size))


class LmDriver:
    """so_icp_lm_* state machine (Ceres restatement) driven from Python -- used by CPU tests."""

    def __init__(self):
        self.L = load(); self.s = LmState()

    @staticmethod
    def sums(cost, count, Jtr, JtJ_full, hist=None):
        s = Sums(); s.cost = cost; s.count = count
        for i in range(6):
            s.Jtr[i] = Jtr[i]
        k = 0
        for i in range(6):
            for j in range(i, 6):
                s.JtJ[k] = JtJ_full[i][j]; k += 1
        if hist is not None:
            for i in range(16):
                s.hist[i] = hist[i]
        return s

    def begin(self, x0, sums, max_iterations=4):
        x0 = np.ascontiguousarray(x0, dtype=np.float64); nxt = np.zeros(7)
        more = self.L.so_icp_lm_begin(C.byref(self.s), _p(x0, C.c_double), C.byref(sums), max_iterations, _p(nxt, C.c_double))
        return more, nxt

    def feed(self, sums):
        nxt = np.zeros(7)
        more = self.L.so_icp_lm_feed(C.byref(self.s), C.byref(sums), _p(nxt, C.c_double))
        return more, nxt

    def result(self):
        pose = np.zeros(7); st = IterStats()
        self.L.so_icp_lm_result(C.byref(self.s), _p(pose, C.c_double), C.byref(st))
        return pose, st
