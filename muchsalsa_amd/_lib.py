"""ctypes binding of include/msgpu.h (muchsalsa_amd/libmsgpu.so).

There is no fallback: if the HIP library is missing or no GPU is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MSGPU_LIB") or os.path.join(_HERE, "libmsgpu.so")  # MSGPU_LIB: another build of the same ABI

ROW_DTYPE = np.dtype([("anchor_id", "<u4"), ("read_id", "<u4"), ("read_len", "<i4"), ("i_lo", "<i4"),
                      ("i_hi", "<i4"), ("n_lo", "<i4"), ("n_hi", "<i4"), ("score", "<u4"), ("line", "<u4"),
                      ("flags", "<u4")])
EDGE_DTYPE = np.dtype([("v1", "<u4"), ("v2", "<u4"), ("em_off", "<u8"), ("order_off", "<u8"), ("em_cnt", "<u4"),
                       ("order_cnt", "<u2"), ("shadow", "u1"), ("pad", "u1")])
EM_DTYPE = np.dtype([("ov_lo", "<i4"), ("ov_hi", "<i4"), ("score", "<f8"), ("anchor_id", "<u4"), ("line", "<u4"),
                     ("flags", "<u4"), ("edge_idx", "<u4")])
ORDER_DTYPE = np.dtype([("edge_idx", "<u4"), ("flags", "<u4"), ("left_offset", "<f8"), ("right_offset", "<f8"),
                        ("score", "<u8"), ("ids_off", "<u8"), ("ids_cnt", "<u4"), ("start", "<u4"), ("end", "<u4"),
                        ("base", "<u4"), ("pad", "<u4", (2,))])
COPY_DTYPE = np.dtype([("src_off", "<u8"), ("dst_off", "<u8"), ("len", "<u4"), ("flags", "<u4")])
assert (ROW_DTYPE.itemsize, EDGE_DTYPE.itemsize, EM_DTYPE.itemsize, ORDER_DTYPE.itemsize) == (40, 32, 32, 64)
assert COPY_DTYPE.itemsize == 24
COPY_ILLUMINA, COPY_REVCOMP = 1, 2
PATH_READ_DTYPE = np.dtype([("read_id", "<u4"), ("direction", "<u4"), ("nanopore_length", "<u8")])
PATH_ORDER_DTYPE = np.dtype([("score", "<u8"), ("base_read", "<u4"), ("ids_off", "<u4"), ("ids_cnt", "<u4"),
                             ("pad", "<u4")])
PATH_EM_DTYPE = np.dtype([("anchor_id", "<u4"), ("ov_lo", "<i4"), ("ov_hi", "<i4")])
PATH_CONTAIN_DTYPE = np.dtype([("host_read", "<u4"), ("nano", "<u4"), ("direction", "<u4"), ("anchors_off", "<u4"),
                               ("anchors_cnt", "<u4")])
PATH_INFO_DTYPE = np.dtype([("target_len", "<u8"), ("target_raw_off", "<u8"), ("query_begin", "<u4"),
                            ("query_end", "<u4"), ("n_anchors", "<u4"), ("n_anchor_edges", "<u4"),
                            ("border_lo", "<i4"), ("border_hi", "<i4"), ("asm_idx", "<i4"), ("pad", "<u4")])
QUERY_INFO_DTYPE = np.dtype([("len", "<u8"), ("raw_off", "<u8"), ("lb", "<i8"), ("rb", "<i8"), ("kind", "<u4"),
                             ("path", "<u4")])
FASTA_RECORD_DTYPE = np.dtype([("raw_off", "<u8"), ("text_off", "<u8"), ("len", "<u4"), ("header_off", "<u4"),
                               ("header_len", "<u4"), ("pad", "<u4")])
assert (PATH_READ_DTYPE.itemsize, PATH_ORDER_DTYPE.itemsize, PATH_EM_DTYPE.itemsize, PATH_CONTAIN_DTYPE.itemsize,
        PATH_INFO_DTYPE.itemsize, QUERY_INFO_DTYPE.itemsize, FASTA_RECORD_DTYPE.itemsize) == (16, 24, 12, 20, 48, 40, 32)
ALIGN_PAIR_DTYPE = np.dtype([("a_off", "<u8"), ("b_off", "<u8"), ("a_len", "<u4"), ("b_len", "<u4")])
assert ALIGN_PAIR_DTYPE.itemsize == 24

OK = 0
E_IO, E_FORMAT, E_NUMBER, E_NOMEM, E_ARG, E_HIP, E_STATE, E_IDS, E_NODEVICE = -1, -2, -3, -4, -5, -6, -7, -8, -9
E_LAYOUT = -10
E_TIMEOUT = -11

ORD_START_V1, ORD_CONTAINED, ORD_DIR, ORD_PRIMARY = 1, 2, 4, 8
BATCH_RESIDENT, BATCH_NO_EDGEMATCHES, BATCH_ROWS_ON_DEVICE, BATCH_ROWS_PACKED = 1, 2, 4, 8


class Params(C.Structure):
    _fields_ = [("min_matches", C.c_uint32), ("th_length", C.c_uint32), ("th_matches", C.c_uint32),
                ("th_overlap", C.c_uint32), ("wiggle_room", C.c_uint64), ("ratio_pct", C.c_double),
                ("alt_frac", C.c_double)]


class PackedRows(C.Structure):  # msgpu_packed_rows: the row table's 28-byte form for the host link
    _fields_ = [("rows", C.c_void_p), ("n_rows", C.c_uint64), ("read_len", C.c_void_p), ("n_reads", C.c_uint32),
                ("n_runs", C.c_uint32), ("run_start", C.c_void_p), ("run_delta", C.c_void_p), ("owner", C.c_void_p)]


class HostTables(C.Structure):  # msgpu_host_tables
    _fields_ = [("edges", C.c_void_p), ("ems", C.c_void_p), ("orders", C.c_void_p), ("ids", C.c_void_p),
                ("read_len", C.c_void_p), ("read_first_line", C.c_void_p), ("n_edges", C.c_uint64),
                ("n_ems", C.c_uint64), ("n_orders", C.c_uint64), ("n_ids", C.c_uint64), ("n_reads", C.c_uint32),
                ("n_anchors", C.c_uint32), ("n_batches", C.c_uint32), ("pad", C.c_uint32), ("wall_ms", C.c_float),
                ("load_ms", C.c_float), ("first_batch_ms", C.c_float), ("compute_done_ms", C.c_float)]


class PathInput(C.Structure):
    _fields_ = [("reads", C.c_void_p), ("n_reads", C.c_uint32), ("asm_idx", C.c_int32), ("order_off", C.c_void_p),
                ("orders", C.c_void_p), ("ids", C.c_void_p), ("em_off", C.c_void_p), ("ems", C.c_void_p),
                ("rows", C.c_void_p), ("n_rows", C.c_size_t), ("contains", C.c_void_p), ("n_contains", C.c_uint32),
                ("pad", C.c_uint32), ("contain_anchors", C.c_void_p)]


class GraphStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_vertices_in", "n_edges_in", "n_contraction_edges", "n_deleted_vertices",
                                          "n_contain_elements", "n_decycled_edges", "n_vertices", "n_edges",
                                          "n_components", "n_paths", "n_path_reads")]


class Counts(C.Structure):
    _fields_ = [("n_rows_in", C.c_uint64), ("n_rows_alive", C.c_uint64), ("n_reads", C.c_uint32),
                ("n_anchors", C.c_uint32), ("n_edges", C.c_uint64), ("n_ems", C.c_uint64), ("n_orders", C.c_uint64),
                ("n_ids", C.c_uint64), ("n_pairs_scanned", C.c_uint64), ("n_edges_fastpath", C.c_uint64),
                ("n_lost_publications", C.c_uint64), ("index_path", C.c_uint64)]


class GroupTables(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("orders", C.c_void_p), ("ids", C.c_void_p), ("read_len", C.c_void_p),
                ("read_first_line", C.c_void_p), ("n_edges", C.c_uint64), ("n_orders", C.c_uint64), ("n_ids", C.c_uint64),
                ("n_ems", C.c_uint64), ("n_reads", C.c_uint32), ("n_anchors", C.c_uint32), ("n_members", C.c_uint32),
                ("id_bytes", C.c_uint32), ("slab_bytes", C.c_uint64), ("wall_ms", C.c_float), ("compute_ms", C.c_float),
                ("exchange_ms", C.c_float), ("rows_sliced", C.c_uint32)]


INDEX_BIN, INDEX_ATOMIC, INDEX_TWO_PASS, INDEX_GENERIC = 0, 1, 2, 4


class Timings(C.Structure):
    _fields_ = [("index_ms", C.c_float), ("candidates_ms", C.c_float), ("chain_ms", C.c_float),
                ("compact_ms", C.c_float), ("chain_kernel_ms", C.c_float), ("chain_kernel_launches", C.c_uint32),
                ("pad", C.c_uint32)]


# every symbol include/msgpu.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("msgpu_default_params", None, [C.POINTER(Params)]),
    ("msgpu_strerror", C.c_char_p, [C.c_int]),
    ("msgpu_create", C.c_int, [C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]),
    ("msgpu_destroy", None, [C.c_void_p]),
    ("msgpu_last_error", C.c_char_p, [C.c_void_p]),
    ("msgpu_set_stream", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msgpu_get_stream", C.c_void_p, [C.c_void_p]),
    ("msgpu_stream_wait", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msgpu_stream_release", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msgpu_set_shard", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("msgpu_set_id_space", C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    ("msgpu_parse_paf", C.c_int, [C.c_char_p, C.POINTER(Params), C.POINTER(C.c_void_p)]),
    ("msgpu_paf_free", None, [C.c_void_p]),
    ("msgpu_paf_rows", C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    ("msgpu_paf_line_count", C.c_size_t, [C.c_void_p]),
    ("msgpu_paf_read_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_paf_anchor_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_paf_read_name", C.c_char_p, [C.c_void_p, C.c_uint32]),
    ("msgpu_paf_anchor_name", C.c_char_p, [C.c_void_p, C.c_uint32]),
    ("msgpu_paf_register_sequences", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("msgpu_load_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_load_rows_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_calculate_edges", C.c_int, [C.c_void_p]),
    ("msgpu_chaining_and_overlaps", C.c_int, [C.c_void_p]),
    ("msgpu_get_counts", C.c_int, [C.c_void_p, C.POINTER(Counts)]),
    ("msgpu_get_timings", C.c_int, [C.c_void_p, C.POINTER(Timings)]),
    ("msgpu_set_stage_events", C.c_int, [C.c_void_p, C.c_int]),
    ("msgpu_copy_tables", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_copy_tables_device", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_copy_reads", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_merge_gathered", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_merge_gathered_ex", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64,
                                          C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    ("msgpu_wire_edges_bytes", C.c_uint64, [C.c_uint64]),
    ("msgpu_wire_orders_bytes", C.c_uint64, [C.c_uint64]),
    ("msgpu_wire_ids_bytes", C.c_uint64, [C.c_uint64, C.c_uint32]),
    ("msgpu_pack_wire", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]),
    ("msgpu_merge_wire", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64,
                                   C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_chain_launches", C.c_uint64, [C.c_void_p]),
    ("msgpu_wait_chain_launch", C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32]),
    ("msgpu_unpack_wire_host", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32]),
    ("msgpu_find_contraction_edges", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_uint32,
                                               C.c_void_p]),
    ("msgpu_pack_rows", C.c_int, [C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(PackedRows)]),
    ("msgpu_packed_rows_free", None, [C.POINTER(PackedRows)]),
    ("msgpu_load_rows_packed", C.c_int, [C.c_void_p, C.POINTER(PackedRows)]),
    ("msgpu_synchronize", C.c_int, [C.c_void_p]),
    ("msgpu_set_deadline", C.c_int, [C.c_void_p, C.c_uint32]),
    ("msgpu_seq_parse", C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    ("msgpu_seq_free", None, [C.c_void_p]),
    ("msgpu_seq_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_seq_name", C.c_char_p, [C.c_void_p, C.c_uint32]),
    ("msgpu_seq_length", C.c_uint64, [C.c_void_p, C.c_uint32]),
    ("msgpu_seq_bases", C.c_void_p, [C.c_void_p, C.c_uint32]),
    ("msgpu_seq_buffer", C.c_void_p, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("msgpu_str_slice", C.c_uint64, [C.c_uint64, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]),
    ("msgpu_seq_create", C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    ("msgpu_seq_destroy", None, [C.c_void_p]),
    ("msgpu_seq_last_error", C.c_char_p, [C.c_void_p]),
    ("msgpu_seq_upload", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]),
    ("msgpu_seq_upload_device", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p,
                                          C.c_uint32]),
    ("msgpu_seq_pack", C.c_int, [C.c_void_p]),
    ("msgpu_seq_pack_store", C.c_int, [C.c_void_p, C.c_int]),
    ("msgpu_seq_upload_bases", C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    ("msgpu_seq_parse_upload", C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_void_p)]),
    ("msgpu_seq_offset", C.c_uint64, [C.c_void_p, C.c_uint32]),
    ("msgpu_seq_set_ids", C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_uint32]),
    ("msgpu_seq_resolve", C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_void_p]),
    ("msgpu_seg_anchor", C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int, C.c_void_p,
                                   C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("msgpu_seg_left_of_anchor", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int,
                                           C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("msgpu_seg_right_of_anchor", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int,
                                            C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64)]),
    ("msgpu_seg_between_anchors", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int, C.c_void_p, C.POINTER(C.c_uint32),
                                            C.POINTER(C.c_int32), C.POINTER(C.c_int)]),
    ("msgpu_consensus_new", C.c_void_p, []),
    ("msgpu_consensus_free", None, [C.c_void_p]),
    ("msgpu_consensus_update", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_int32, C.c_int32]),
    ("msgpu_consensus_borders", C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                          C.POINTER(C.c_uint64)]),
    ("msgpu_consensus_pieces", C.c_size_t, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t]),
    ("msgpu_consensus_clone", C.c_void_p, [C.c_void_p]),
    ("msgpu_assembly_create", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    ("msgpu_assembly_free", None, [C.c_void_p]),
    ("msgpu_assembly_last_error", C.c_char_p, [C.c_void_p]),
    ("msgpu_assembly_set_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_assembly_borrow_rows", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_assembly_add_path", C.c_int, [C.c_void_p, C.POINTER(PathInput)]),
    ("msgpu_assembly_add_paths", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    ("msgpu_assembly_add_graph_paths", C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]),
    ("msgpu_assembly_path_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_assembly_query_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_assembly_path_info", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    ("msgpu_assembly_query_info", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p]),
    ("msgpu_assembly_pieces", C.c_size_t, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_assembly_raw_bytes", C.c_uint64, [C.c_void_p]),
    ("msgpu_assembly_finish", C.c_int, [C.c_void_p, C.c_void_p]),
    ("msgpu_assembly_validate", C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.POINTER(C.c_uint64)]),
    ("msgpu_assembly_text", C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    ("msgpu_fasta_text_bytes", C.c_uint64, [C.c_uint32, C.c_uint64]),
    ("msgpu_fasta_format", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p, C.c_size_t,
                                     C.c_void_p, C.c_uint64, C.c_void_p]),
    ("msgpu_graph_create", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p,
                                     C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p)]),
    ("msgpu_graph_create_borrowed", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64,
                                              C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_uint32,
                                              C.POINTER(C.c_void_p)]),
    ("msgpu_graph_path_edges", C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    ("msgpu_graph_set_path_edgematches", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_graph_free", None, [C.c_void_p]),
    ("msgpu_graph_last_error", C.c_char_p, [C.c_void_p]),
    ("msgpu_graph_clean_up", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    ("msgpu_graph_set_threads", C.c_int, [C.c_void_p, C.c_uint32]),
    ("msgpu_graph_linearize", C.c_int, [C.c_void_p]),
    ("msgpu_graph_get_stats", C.c_int, [C.c_void_p, C.POINTER(GraphStats)]),
    ("msgpu_graph_path_count", C.c_uint32, [C.c_void_p]),
    ("msgpu_graph_path_input", C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(PathInput)]),
    ("msgpu_graph_state", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("msgpu_graph_max_span_tree", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                            C.c_void_p]),
    ("msgpu_graph_bookkeeping", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_void_p, C.c_size_t,
                                          C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("msgpu_graph_connected_components", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64,
                                                   C.c_void_p, C.POINTER(C.c_uint32)]),
    ("msgpu_graph_shortest_path", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_uint32,
                                            C.c_uint32, C.c_void_p, C.POINTER(C.c_uint32)]),
    ("msgpu_graph_sort_topologically", C.c_int, [C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p,
                                                 C.POINTER(C.c_uint32)]),
    ("msgpu_group_create", C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p)]),
    ("msgpu_group_destroy", None, [C.c_void_p]),
    ("msgpu_group_last_error", C.c_char_p, [C.c_void_p]),
    ("msgpu_group_size", C.c_int, [C.c_void_p]),
    ("msgpu_group_ctx", C.c_void_p, [C.c_void_p, C.c_int]),
    ("msgpu_group_overlap", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(GroupTables)]),
    ("msgpu_group_set_timeout", C.c_int, [C.c_void_p, C.c_uint32]),
    ("msgpu_group_device_tables", C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                            C.POINTER(C.c_void_p)]),
    ("msgpu_overlap_batched", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.POINTER(HostTables)]),
    ("msgpu_overlap_batched_ex", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32,
                                           C.POINTER(HostTables)]),
    ("msgpu_get_edgematches", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("msgpu_pinned_alloc", C.c_void_p, [C.c_size_t]),
    ("msgpu_pinned_free", None, [C.c_void_p]),
    ("msgpu_index_lines", C.c_int, [C.c_char_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    ("msgpu_registry_new", C.c_void_p, []),
    ("msgpu_registry_free", None, [C.c_void_p]),
    ("msgpu_registry_id", C.c_uint32, [C.c_void_p, C.c_char_p]),
    ("msgpu_registry_size", C.c_uint32, [C.c_void_p]),
    ("msgpu_registry_clear", None, [C.c_void_p]),
    ("msgpu_toggle_mul", C.c_int, [C.c_int, C.c_int]),
    ("msgpu_gather_plan_create", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    ("msgpu_gather_plan_free", None, [C.c_void_p]),
    ("msgpu_gather_plan_out_bytes", C.c_uint64, [C.c_void_p]),
    ("msgpu_gather_plan_bases", C.c_uint64, [C.c_void_p]),
    ("msgpu_gather_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    ("msgpu_seq_synchronize", C.c_int, [C.c_void_p]),
    ("msgpu_edit_distance", C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32,
                                      C.c_void_p]),
]

_lib = None
PRELOAD_TORCH = True  # the command-line driver turns this off: it never imports torch, so there is nothing to reconcile


def lib():
    """Load libmsgpu.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "muchsalsa_amd/libmsgpu.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C muchsalsa_amd/csrc`). There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own HIP/HSA runtime.  Two HIP runtimes in one process do not share the
        # device: whichever is initialised second sees "no GPU".  Importing torch first (when it is installed) makes
        # libmsgpu's libamdhip64.so.7 dependency resolve to the copy torch already loaded, so load order stops mattering.
        import sys
        if PRELOAD_TORCH or "torch" in sys.modules:
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        handle = C.CDLL(LIB_PATH)
        for name, restype, argtypes in SYMBOLS:
            fn = getattr(handle, name)  # AttributeError if the library lacks a declared symbol
            fn.restype = restype
            fn.argtypes = argtypes
        _lib = handle
    return _lib
