//! Raw bindings to `include/frieda_hip.h` (ABI version 1) and a minimal safe layer.
//!
//! Status codes: 0 ok, 1 argument, 2 HIP, 3 invariant (= the reference's `panic!`), 4 out of memory, 5 malformed image.
#![allow(non_camel_case_types)]

use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
pub struct frieda_ctx {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frieda_multi {
    _private: [u8; 0],
}
#[repr(C)]
pub struct frieda_proof {
    _private: [u8; 0],
}

/// stwo `PcsConfig { pow_bits, fri_config: FriConfig { log_blowup_factor, log_last_layer_degree_bound, n_queries } }`
#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct frieda_pcs_config {
    pub pow_bits: u32,
    pub log_blowup_factor: u32,
    pub log_last_layer_degree_bound: u32,
    pub n_queries: u32,
}

pub const FRIEDA_OK: c_int = 0;
pub const FRIEDA_ERR_ARG: c_int = 1;
pub const FRIEDA_ERR_HIP: c_int = 2;
pub const FRIEDA_ERR_INVARIANT: c_int = 3;
pub const FRIEDA_ERR_NOMEM: c_int = 4;
pub const FRIEDA_ERR_FORMAT: c_int = 5;

extern "C" {
    pub fn frieda_abi_version() -> u32;
    pub fn frieda_status_string(status: c_int) -> *const c_char;
    pub fn frieda_last_error(ctx: *const frieda_ctx) -> *const c_char;
    pub fn frieda_ctx_notes(ctx: *const frieda_ctx) -> *const c_char;

    // context
    pub fn frieda_ctx_create(device: c_int, stream: *mut c_void, out: *mut *mut frieda_ctx) -> c_int;
    pub fn frieda_ctx_destroy(ctx: *mut frieda_ctx) -> c_int;
    pub fn frieda_ctx_synchronize(ctx: *mut frieda_ctx) -> c_int;
    pub fn frieda_ctx_release_workspace(ctx: *mut frieda_ctx) -> c_int;
    pub fn frieda_ctx_set_twiddle_cache(ctx: *mut frieda_ctx, enabled: c_int) -> c_int;
    pub fn frieda_ctx_set_host_channel(ctx: *mut frieda_ctx, enabled: c_int) -> c_int;
    /// per-context tuning / A-B option, named like the environment variable that sets its default
    pub fn frieda_ctx_set_option(ctx: *mut frieda_ctx, name: *const c_char, value: i64) -> c_int;
    /// test hook (include/frieda_hip_testing.h), not part of the boundary
    pub fn frieda_ctx_test_set_draw_bound(ctx: *mut frieda_ctx, bound: u32) -> c_int;
    pub fn frieda_ctx_test_set_grind_first_log(ctx: *mut frieda_ctx, log_first: u32) -> c_int;
    pub fn frieda_ctx_test_set_arena_limit(ctx: *mut frieda_ctx, bytes: u64) -> c_int;
    pub fn frieda_test_near_cpus(sysfs_root: *const c_char, pci_bus_id: *const c_char, out_cpus: *mut c_int, cap: usize, n: *mut usize) -> c_int;
    pub fn frieda_test_parse_cpulist(text: *const c_char, out_cpus: *mut c_int, cap: usize, n: *mut usize) -> c_int;
    /// batch policy: device workspace one blob adds to a batched call; the cut of `count` equal-length blobs into calls
    pub fn frieda_workspace_bytes(len: usize, log_blowup_factor: u32, log_last_layer_degree_bound: u32, prove: c_int) -> usize;
    pub fn frieda_batch_plan(ctx: *const frieda_ctx, len: usize, log_blowup_factor: u32, log_last_layer_degree_bound: u32, prove: c_int, count: u32, in_flight: u32, out_calls: *mut u32, cap: usize, n_calls: *mut u32) -> c_int;
    pub fn frieda_ctx_set_kernel_timing(ctx: *mut frieda_ctx, enabled: c_int) -> c_int;
    pub fn frieda_ctx_last_prove_phases(ctx: *const frieda_ctx, out_ms: *mut f64) -> c_int;
    /// measurement aid: pure-compute Blake2s compression rate of this device (leaf-shaped, node-shaped), compressions per second
    pub fn frieda_ctx_blake2s_ceiling(ctx: *mut frieda_ctx, leaf_per_s: *mut f64, node_per_s: *mut f64) -> c_int;
    pub fn frieda_ctx_blake2s_ceiling_ex(ctx: *mut frieda_ctx, out: *mut f64) -> c_int;
    /// diagnostic: alphas per FRI layer and the pre-grind channel digest of the last finished proof
    pub fn frieda_ctx_last_transcript(ctx: *const frieda_ctx, n_layers: *mut u32, alphas: *mut u32, cap_layers: usize, digest_before_grind: *mut u8) -> c_int;
    pub fn frieda_ctx_kernel_timing_report(ctx: *mut frieda_ctx, buf: *mut c_char, cap: usize, reset: c_int) -> usize;

    // Level A
    pub fn frieda_commit(ctx: *mut frieda_ctx, data: *const u8, len: usize, log_blowup_factor: u32, out_root: *mut u8) -> c_int;
    pub fn frieda_commit_device(ctx: *mut frieda_ctx, d_data: *const c_void, len: usize, log_blowup_factor: u32, d_out_root: *mut c_void) -> c_int;
    pub fn frieda_commit_and_generate_proof(ctx: *mut frieda_ctx, data: *const u8, len: usize, seed: *const u64, cfg: frieda_pcs_config, out_commitment: *mut u8, out: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_commit_and_generate_proof_device(ctx: *mut frieda_ctx, d_data: *const c_void, len: usize, seed: *const u64, cfg: frieda_pcs_config, out_commitment: *mut u8, out: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_prove_begin(ctx: *mut frieda_ctx, data: *const u8, len: usize, seed: *const u64, cfg: frieda_pcs_config) -> c_int;
    pub fn frieda_prove_begin_device(ctx: *mut frieda_ctx, d_data: *const c_void, len: usize, seed: *const u64, cfg: frieda_pcs_config) -> c_int;
    pub fn frieda_prove_finish(ctx: *mut frieda_ctx, out_commitment: *mut u8, out: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_commit_and_generate_proof_batch(ctx: *mut frieda_ctx, data: *const u8, stride: usize, len: usize, count: u32, seeds: *const u64, cfg: frieda_pcs_config, out_commitments: *mut u8, out_proofs: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_commit_and_generate_proof_batch_device(ctx: *mut frieda_ctx, d_data: *const c_void, stride: usize, len: usize, count: u32, seeds: *const u64, cfg: frieda_pcs_config, out_commitments: *mut u8, out_proofs: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_prove_batch_begin(ctx: *mut frieda_ctx, data: *const u8, stride: usize, len: usize, count: u32, seeds: *const u64, cfg: frieda_pcs_config) -> c_int;
    pub fn frieda_prove_batch_begin_device(ctx: *mut frieda_ctx, d_data: *const c_void, stride: usize, len: usize, count: u32, seeds: *const u64, cfg: frieda_pcs_config) -> c_int;
    pub fn frieda_prove_batch_finish(ctx: *mut frieda_ctx, count: u32, out_commitments: *mut u8, out_proofs: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_commit_batch(ctx: *mut frieda_ctx, data: *const u8, stride: usize, len: usize, count: u32, log_blowup_factor: u32, out_roots: *mut u8) -> c_int;
    pub fn frieda_commit_batch_device(ctx: *mut frieda_ctx, d_data: *const c_void, stride: usize, len: usize, count: u32, log_blowup_factor: u32, out_roots: *mut u8) -> c_int;
    pub fn frieda_generate_proof(ctx: *mut frieda_ctx, data: *const u8, len: usize, seed: *const u64, cfg: frieda_pcs_config, out: *mut *mut frieda_proof) -> c_int;
    // multi-GPU: blob i -> devices[i mod n]; roots gathered with ncclAllGather on a single-process communicator
    pub fn frieda_multi_create(devices: *const c_int, n_devices: u32, out: *mut *mut frieda_multi) -> c_int;
    pub fn frieda_multi_destroy(m: *mut frieda_multi) -> c_int;
    pub fn frieda_multi_device_count(m: *const frieda_multi) -> u32;
    pub fn frieda_multi_last_error(m: *const frieda_multi) -> *const c_char;
    pub fn frieda_multi_uses_rccl(m: *const frieda_multi) -> c_int;
    pub fn frieda_multi_gather_count(m: *const frieda_multi) -> u64;
    pub fn frieda_multi_ctx(m: *mut frieda_multi, device_slot: u32) -> *mut frieda_ctx;
    pub fn frieda_multi_release_workspace(m: *mut frieda_multi) -> c_int;
    pub fn frieda_multi_near_cpus(m: *const frieda_multi, device_slot: u32, out_cpus: *mut c_int, cap: usize) -> u32;
    pub fn frieda_commit_many(m: *mut frieda_multi, blobs: *const *const u8, lens: *const usize, count: u32, log_blowup_factor: u32, out_roots: *mut u8) -> c_int;
    pub fn frieda_prove_many(m: *mut frieda_multi, blobs: *const *const u8, lens: *const usize, count: u32, seeds: *const u64, cfg: frieda_pcs_config, out_commitments: *mut u8, out_proofs: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_verify(proof: *const frieda_proof, seed: *const u64, ok: *mut c_int) -> c_int;
    /// verify + the positions the accepted proof sampled (evaluations[i] sits at out_positions[i] of the bit-reversed codeword)
    pub fn frieda_verify_samples(proof: *const frieda_proof, seed: *const u64, ok: *mut c_int, out_positions: *mut u32, cap: usize, n_positions: *mut usize) -> c_int;

    // struct Proof
    pub fn frieda_proof_free(p: *mut frieda_proof);
    pub fn frieda_proof_clone(p: *const frieda_proof, out: *mut *mut frieda_proof) -> c_int;
    pub fn frieda_proof_proof_of_work(p: *const frieda_proof) -> u64;
    pub fn frieda_proof_set_proof_of_work(p: *mut frieda_proof, nonce: u64);
    pub fn frieda_proof_pcs_config(p: *const frieda_proof) -> frieda_pcs_config;
    pub fn frieda_proof_log_size_bound(p: *const frieda_proof) -> u32;
    pub fn frieda_proof_n_evaluations(p: *const frieda_proof) -> usize;
    pub fn frieda_proof_evaluations(p: *mut frieda_proof) -> *mut u32;
    pub fn frieda_proof_resize_evaluations(p: *mut frieda_proof, n: usize) -> c_int;
    pub fn frieda_proof_n_inner_layers(p: *const frieda_proof) -> usize;
    pub fn frieda_proof_layer_commitment(p: *const frieda_proof, layer: usize) -> *const u8;
    pub fn frieda_proof_layer_fri_witness(p: *const frieda_proof, layer: usize, n_qm31: *mut usize) -> *const u32;
    pub fn frieda_proof_layer_hash_witness(p: *const frieda_proof, layer: usize, n_hashes: *mut usize) -> *const u8;
    pub fn frieda_proof_layer_column_witness(p: *const frieda_proof, layer: usize, n_m31: *mut usize) -> *const u32;
    pub fn frieda_proof_last_layer_poly(p: *const frieda_proof, n_qm31: *mut usize) -> *const u32;
    pub fn frieda_proof_serialize(p: *const frieda_proof, buf: *mut u8, cap: usize) -> usize;
    pub fn frieda_proof_deserialize(buf: *const u8, len: usize, out: *mut *mut frieda_proof) -> c_int;

    // Level B: Column storage
    pub fn frieda_dev_alloc(ctx: *mut frieda_ctx, bytes: usize, d_out: *mut *mut c_void) -> c_int;
    pub fn frieda_dev_free(ctx: *mut frieda_ctx, d: *mut c_void) -> c_int;
    pub fn frieda_dev_upload(ctx: *mut frieda_ctx, d_dst: *mut c_void, h_src: *const c_void, bytes: usize) -> c_int;
    pub fn frieda_dev_download(ctx: *mut frieda_ctx, h_dst: *mut c_void, d_src: *const c_void, bytes: usize) -> c_int;
    // Level B: codec, PolyOps, MerkleOps, FriOps, GrindOps
    /// Column::at for a BaseField column / a SecureColumn (SoA); synchronises the ctx stream
    pub fn frieda_dev_at(ctx: *mut frieda_ctx, d_col: *const u32, index: usize, out: *mut u32) -> c_int;
    pub fn frieda_dev_at_secure(ctx: *mut frieda_ctx, d_cols: *const u32, stride: usize, index: usize, out: *mut u32) -> c_int;
    /// ColumnOps::bit_reverse_column, in place (ncols = 1: BaseField column; 4: SecureColumn)
    pub fn frieda_bit_reverse_column(ctx: *mut frieda_ctx, d_cols: *mut u32, stride: usize, ncols: u32, log_size: u32) -> c_int;
    pub fn frieda_codec_shape(len: usize, n_felts: *mut usize, n_padded: *mut usize, log_size: *mut u32) -> c_int;
    pub fn frieda_unpack30(ctx: *mut frieda_ctx, d_bytes: *const c_void, len: usize, d_coef: *mut u32, n_out: usize) -> c_int;
    pub fn frieda_pack30(ctx: *mut frieda_ctx, d_felts: *const u32, n_felts: usize, d_bytes: *mut c_void, len: usize) -> c_int;
    pub fn frieda_precompute_twiddles(ctx: *mut frieda_ctx, log_domain: u32, d_twiddles: *mut *const u32, d_inv_twiddles: *mut *const u32) -> c_int;
    pub fn frieda_circle_evaluate(ctx: *mut frieda_ctx, d_coef: *const u32, ncols: u32, log_coef: u32, log_domain: u32, d_out: *mut u32) -> c_int;
    pub fn frieda_circle_interpolate(ctx: *mut frieda_ctx, d_block: *const u32, ncols: u32, log_coef: u32, log_domain: u32, block: u32, d_coef: *mut u32) -> c_int;
    pub fn frieda_reconstruct_device(ctx: *mut frieda_ctx, d_block: *const u32, log_coef: u32, log_domain: u32, block: u32, len: usize, d_out_bytes: *mut c_void) -> c_int;
    pub fn frieda_circle_interpolate_cells(ctx: *mut frieda_ctx, d_cells: *const u32, cell_index: *const u32, n_cells: u32, ncols: u32, log_cell: u32, log_coef: u32, log_domain: u32, d_coef: *mut u32) -> c_int;
    /// over-determined form: picks an independent subset of the offered cells (at most 256 needed)
    pub fn frieda_circle_interpolate_cells_any(ctx: *mut frieda_ctx, d_cells: *const u32, cell_index: *const u32, n_avail: u32, ncols: u32, log_cell: u32, log_coef: u32, log_domain: u32, d_coef: *mut u32, out_used: *mut u32) -> c_int;
    pub fn frieda_reconstruct_cells_device(ctx: *mut frieda_ctx, d_cells: *const u32, cell_index: *const u32, n_cells: u32, log_cell: u32, log_coef: u32, log_domain: u32, len: usize, d_out_bytes: *mut c_void) -> c_int;
    /// any >= 2^log_coef + 2 sampled points, no bound on their number (erasure-locator route, no linear system)
    pub fn frieda_circle_interpolate_points(ctx: *mut frieda_ctx, d_cells: *const u32, cell_index: *const u32, n_cells: u32, ncols: u32, log_cell: u32, log_coef: u32, log_domain: u32, d_coef: *mut u32) -> c_int;
    pub fn frieda_reconstruct_points_device(ctx: *mut frieda_ctx, d_cells: *const u32, cell_index: *const u32, n_cells: u32, log_cell: u32, log_coef: u32, log_domain: u32, len: usize, d_out_bytes: *mut c_void) -> c_int;
    pub fn frieda_merkle_commit_layer(ctx: *mut frieda_ctx, log_size: u32, d_prev: *const c_void, d_cols: *const *const u32, ncols: u32, d_out: *mut c_void) -> c_int;
    pub fn frieda_merkle_commit(ctx: *mut frieda_ctx, d_cols: *const u32, log_size: u32, d_layers: *mut c_void) -> c_int;
    pub fn frieda_merkle_layer_offset(log_size: u32, layer_log: u32) -> usize;
    pub fn frieda_merkle_root(ctx: *mut frieda_ctx, d_cols: *const u32, log_size: u32, d_root: *mut c_void) -> c_int;
    pub fn frieda_fold_circle_into_line(ctx: *mut frieda_ctx, d_dst: *mut u32, d_src: *const u32, log_domain: u32, alpha: *const u32) -> c_int;
    pub fn frieda_fold_line(ctx: *mut frieda_ctx, d_src: *const u32, line_log: u32, log_domain: u32, alpha: *const u32, d_dst: *mut u32) -> c_int;
    /// evaluate + fold_circle_into_line + one fold_line in one pass over the evaluation (both challenges known to the caller)
    pub fn frieda_circle_evaluate_fold2(ctx: *mut frieda_ctx, d_coeffs: *const u32, log_size: u32, log_domain: u32, d_evals: *mut u32, alpha0: *const u32, accumulate_line1: c_int, d_line1: *mut u32, alpha1: *const u32, d_line2: *mut u32) -> c_int;
    pub fn frieda_circle_extend(ctx: *mut frieda_ctx, d_coef: *const u32, ncols: u32, log_coef: u32, log_size: u32, d_out: *mut u32) -> c_int;
    pub fn frieda_circle_eval_at_point(ctx: *mut frieda_ctx, d_coef: *const u32, ncols: u32, log_coef: u32, point_x: *const u32, point_y: *const u32, out: *mut u32) -> c_int;
    pub fn frieda_fri_decompose(ctx: *mut frieda_ctx, d_eval: *const u32, log_size: u32, d_g: *mut u32, out_lambda: *mut u32) -> c_int;
    pub fn frieda_grind(ctx: *mut frieda_ctx, digest: *const u8, pow_bits: u32, nonce: *mut u64) -> c_int;
}

// ---- minimal safe layer: what a frieda fork's src/commit.rs and src/proof.rs forward to -------------------------------

/// One device + one stream + workspace.  `!Sync`: use one per thread / per GPU.
pub struct Context(*mut frieda_ctx);

fn check(rc: c_int) {
    // FRIEDA_ERR_INVARIANT is exactly where the reference panics (assert!/unwrap); every other failure is fatal too.
    if rc != FRIEDA_OK {
        panic!("frieda_hip: status {rc}");
    }
}

impl Context {
    pub fn new(device: i32) -> Self {
        let mut c = core::ptr::null_mut();
        check(unsafe { frieda_ctx_create(device, core::ptr::null_mut(), &mut c) });
        Context(c)
    }
    /// `frieda::api::commit` (src/lib.rs:31)
    pub fn commit(&mut self, data: &[u8], log_blowup_factor: u32) -> [u8; 32] {
        let mut root = [0u8; 32];
        check(unsafe { frieda_commit(self.0, data.as_ptr(), data.len(), log_blowup_factor, root.as_mut_ptr()) });
        root
    }
    /// `frieda::proof::commit_and_generate_proof` (src/proof.rs:32); the proof comes back as an owned handle
    pub fn commit_and_generate_proof(&mut self, data: &[u8], seed: Option<u64>, cfg: frieda_pcs_config) -> ([u8; 32], Proof) {
        let mut root = [0u8; 32];
        let mut p = core::ptr::null_mut();
        let s = seed.unwrap_or(0);
        let sp = if seed.is_some() { &s as *const u64 } else { core::ptr::null() };
        check(unsafe { frieda_commit_and_generate_proof(self.0, data.as_ptr(), data.len(), sp, cfg, root.as_mut_ptr(), &mut p) });
        (root, Proof(p))
    }
}
impl Drop for Context {
    fn drop(&mut self) {
        unsafe { frieda_ctx_destroy(self.0) };
    }
}

pub struct Proof(*mut frieda_proof);
impl Proof {
    /// `frieda::api::verify` (src/lib.rs:41): `false` on rejection, panic where the reference panics
    pub fn verify(&self, seed: Option<u64>) -> bool {
        let mut ok: c_int = 0;
        let s = seed.unwrap_or(0);
        let sp = if seed.is_some() { &s as *const u64 } else { core::ptr::null() };
        check(unsafe { frieda_verify(self.0, sp, &mut ok) });
        ok != 0
    }
    pub fn proof_of_work(&self) -> u64 {
        unsafe { frieda_proof_proof_of_work(self.0) }
    }
    /// evaluations as `[u32; 4]` QM31 coordinates (stwo `QM31::from_u32_unchecked(a, b, c, d)`)
    pub fn evaluations(&self) -> Vec<[u32; 4]> {
        let n = unsafe { frieda_proof_n_evaluations(self.0) };
        let p = unsafe { frieda_proof_evaluations(self.0) };
        (0..n).map(|i| unsafe { [*p.add(4 * i), *p.add(4 * i + 1), *p.add(4 * i + 2), *p.add(4 * i + 3)] }).collect()
    }
    pub fn to_bytes(&self) -> Vec<u8> {
        let n = unsafe { frieda_proof_serialize(self.0, core::ptr::null_mut(), 0) };
        let mut b = vec![0u8; n];
        unsafe { frieda_proof_serialize(self.0, b.as_mut_ptr(), n) };
        b
    }
}
impl Drop for Proof {
    fn drop(&mut self) {
        unsafe { frieda_proof_free(self.0) };
    }
}
