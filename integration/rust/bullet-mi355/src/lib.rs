//! Rust binding of `libbullet_amd.so` (C ABI: `include/bullet_amd.h`).
//!
//! Two layers: [`ffi`] = the raw declarations, one per entry point a bullet backend needs (same order as the header);
//! and safe wrappers ([`Net`], [`Reader`], [`train`]) shaped so that `bullet_trainer::run::train`
//! (`crates/trainer/src/run.rs:75-265`) and `ValueTrainer::run` (`crates/bullet_lib/src/value.rs:141-211`) can call them
//! behind a `mi355` cargo feature: `Function::execute` + `Optimiser::update` become [`Net::forward_backward`] +
//! [`Net::optimiser_step`], or the whole loop becomes [`train`].
//!
//! Not compiled in the repository this file ships in (no Rust toolchain there); the `#[repr(C)]` layouts and
//! prototypes are checked against the header by `tests/test_rust_binding_layout.py`.

use std::ffi::{c_char, c_int, c_void, CStr, CString};

pub mod ffi {
    use std::ffi::{c_char, c_int, c_void};

    pub const BT_MAX_DENSE: usize = 4;

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct bt_net_desc {
        pub num_inputs: c_int,
        pub nnz: c_int,
        pub ft_out: c_int,
        pub ft_act: c_int,
        pub pairwise: c_int,
        pub perspectives: c_int,
        pub num_buckets: c_int,
        pub n_dense: c_int,
        pub dense_out: [c_int; BT_MAX_DENSE],
        pub dense_act: [c_int; BT_MAX_DENSE],
        pub dense_bucketed: [c_int; BT_MAX_DENSE],
        pub use_entry_weights: c_int,
        pub ft_factoriser: c_int,
        pub dense_dual: [c_int; BT_MAX_DENSE],
        pub dense_skip: [c_int; BT_MAX_DENSE],
        pub ft_segment: c_int,
        pub loss_kind: c_int,
        pub loss_power: f32,
    }

    #[repr(C)]
    pub struct bt_net {
        _private: [u8; 0],
    }
    #[repr(C)]
    pub struct bt_reader {
        _private: [u8; 0],
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct bt_batch {
        pub batch: c_int,
        pub stm: *const i32,
        pub nstm: *const i32,
        pub buckets: *const i32,
        pub targets: *const f32,
        pub entry_weights: *const f32,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct bt_input_desc {
        pub kind: c_int,
        pub factorised: c_int,
        pub buckets: [c_int; 64],
        pub num_buckets: c_int,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct bt_map_opts {
        pub out_buckets: c_int,
        pub blend: f32,
        pub eval_scale: f32,
        pub use_win_rate_model: c_int,
        pub wdl_output: c_int,
        pub threads: c_int,
        pub blend_fn: Option<unsafe extern "C" fn(user: *mut c_void, record32: *const c_void, blend: f32) -> f32>,
        pub weight_fn: Option<unsafe extern "C" fn(user: *mut c_void, record32: *const c_void) -> f32>,
        pub fn_user: *mut c_void,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct bt_lr_sched {
        pub kind: c_int,
        pub a: f32,
        pub b: f32,
        pub n: c_int,
        pub warmup_batches: c_int,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct bt_wdl_sched {
        pub kind: c_int,
        pub start: f32,
        pub end: f32,
        pub warmup_batches: c_int,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct bt_train_config {
        pub batch_size: c_int,
        pub batches_per_superbatch: c_int,
        pub start_superbatch: c_int,
        pub end_superbatch: c_int,
        pub lr: bt_lr_sched,
        pub wdl: bt_wdl_sched,
        pub eval_scale: f32,
        pub out_buckets: c_int,
        pub use_win_rate_model: c_int,
        pub threads: c_int,
        pub log_rate: c_int,
        pub wdl_output: c_int,
        pub device_mapper: c_int,
        pub dp_rank: c_int,
        pub dp_world: c_int,
        pub grad_sync:
            Option<unsafe extern "C" fn(user: *mut c_void, buf_dev: *mut f32, n_floats: usize, stream: *mut c_void) -> c_int>,
        pub grad_sync_user: *mut c_void,
        pub dp_loss_slot: c_int,
        pub stop_superbatch: c_int,
        pub reader: *mut c_void,
        pub lr_fn: Option<unsafe extern "C" fn(user: *mut c_void, batch: c_int, superbatch: c_int) -> f32>,
        pub wdl_fn: Option<unsafe extern "C" fn(user: *mut c_void, batch: c_int, superbatch: c_int, max: c_int) -> f32>,
        pub sched_user: *mut c_void,
        pub blend_fn: Option<unsafe extern "C" fn(user: *mut c_void, record32: *const c_void, blend: f32) -> f32>,
        pub weight_fn: Option<unsafe extern "C" fn(user: *mut c_void, record32: *const c_void) -> f32>,
        pub fn_user: *mut c_void,
        pub comm: *mut c_void,
        pub dp_shard_min_floats: usize,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct bt_comm_hooks {
        pub all_reduce: Option<unsafe extern "C" fn(user: *mut c_void, buf_dev: *mut f32, n: usize, stream: *mut c_void) -> c_int>,
        pub reduce_scatter: Option<
            unsafe extern "C" fn(user: *mut c_void, send_dev: *const f32, recv_dev: *mut f32, n_per_rank: usize, stream: *mut c_void) -> c_int,
        >,
        pub all_gather: Option<
            unsafe extern "C" fn(user: *mut c_void, send_dev: *const f32, recv_dev: *mut f32, n_per_rank: usize, stream: *mut c_void) -> c_int,
        >,
        pub user: *mut c_void,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug, Default)]
    pub struct bt_dp_plan {
        pub sharded: c_int,
        pub table_n: usize,
        pub per_rank: usize,
        pub slice_lo: usize,
        pub tail_off: usize,
        pub tail_n: usize,
    }

    #[repr(C)]
    #[derive(Clone, Copy)]
    pub struct bt_saved_format {
        pub id: *const c_char,
        pub custom: *const c_void,
        pub custom_len: usize,
        pub quant: c_int,
        pub multiplier: c_int,
        pub round: c_int,
        pub transpose: c_int,
        pub merge_factoriser: c_int,
        pub transform:
            Option<unsafe extern "C" fn(user: *mut c_void, id: *const c_char, values: *mut f32, count: usize) -> c_int>,
        pub user: *mut c_void,
    }

    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct bt_sf_entry {
        pub chessboard: *const c_void,
        pub score: i16,
        pub ply: u16,
        pub result: i16,
        pub move_from: u8,
        pub move_to: u8,
        pub move_type: u8,
        pub promotion: u8,
        pub captured_piece: u8,
        pub in_check: u8,
    }

    pub type bt_batch_callback = Option<unsafe extern "C" fn(user: *mut c_void, superbatch: c_int, batch: c_int, loss: f32)>;
    pub type bt_superbatch_callback =
        Option<unsafe extern "C" fn(user: *mut c_void, superbatch: c_int, running_loss: f32, positions_per_sec: f32)>;
    pub type bt_viri_filter = Option<
        unsafe extern "C" fn(user: *mut c_void, chessboard32: *const c_void, mv: u16, eval: i16, wdl: f32, ply: c_int) -> c_int,
    >;
    pub type bt_sf_filter = Option<unsafe extern "C" fn(user: *mut c_void, entry: *const bt_sf_entry) -> c_int>;
    pub type bt_monty_filter =
        Option<unsafe extern "C" fn(user: *mut c_void, chessboard32: *const c_void, mv: u16, score: i16, result: f32) -> c_int>;

    /// `viriformat::dataformat::Filter` as the library restates it (`ViriFilter::Builtin`)
    #[repr(C)]
    #[derive(Clone, Copy, Debug)]
    pub struct bt_viri_builtin_filter {
        pub min_ply: u32,
        pub min_pieces: u32,
        pub max_eval: u32,
        pub filter_tactical: c_int,
        pub filter_check: c_int,
        pub filter_castling: c_int,
        pub max_eval_incorrectness: u32,
        pub random_fen_skipping: c_int,
        pub random_fen_skip_probability: f64,
    }

    extern "C" {
        pub fn bt_last_error() -> *const c_char;
        pub fn bt_device_set(ordinal: c_int) -> c_int;
        pub fn bt_device_sync() -> c_int;
        pub fn bt_malloc(ptr: *mut *mut c_void, bytes: usize) -> c_int;
        pub fn bt_free(ptr: *mut c_void) -> c_int;
        pub fn bt_stream_create(stream: *mut *mut c_void) -> c_int;
        pub fn bt_stream_destroy(stream: *mut c_void) -> c_int;
        pub fn bt_stream_sync(stream: *mut c_void) -> c_int;
        pub fn bt_memcpy_h2d_async(dst: *mut c_void, src_host: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;
        pub fn bt_memcpy_d2h_async(dst_host: *mut c_void, src: *const c_void, bytes: usize, stream: *mut c_void) -> c_int;

        pub fn bt_net_param_count(desc: *const bt_net_desc) -> usize;
        pub fn bt_net_num_tensors(desc: *const bt_net_desc) -> c_int;
        pub fn bt_net_tensor_size(desc: *const bt_net_desc, tensor: c_int) -> usize;
        pub fn bt_net_tensor_offset(desc: *const bt_net_desc, tensor: c_int) -> usize;
        pub fn bt_net_tensor_name(desc: *const bt_net_desc, tensor: c_int) -> *const c_char;
        pub fn bt_net_create(
            desc: *const bt_net_desc, max_batch: c_int, ext_params: *mut f32, ext_grads: *mut f32, net: *mut *mut bt_net,
        ) -> c_int;
        pub fn bt_net_destroy(net: *mut bt_net) -> c_int;
        pub fn bt_net_params(net: *mut bt_net) -> *mut f32;
        pub fn bt_net_grads(net: *mut bt_net) -> *mut f32;
        pub fn bt_net_set_adamw(
            net: *mut bt_net, tensor: c_int, decay: f32, beta1: f32, beta2: f32, min_w: f32, max_w: f32,
        ) -> c_int;
        pub fn bt_net_set_optimiser(net: *mut bt_net, kind: c_int, n_sma_threshold: f32, alpha: f32, k: c_int) -> c_int;
        pub fn bt_net_forward_backward(
            net: *mut bt_net, stream: *mut c_void, batch: *const bt_batch, loss_sum_dev: *mut f32,
        ) -> c_int;
        pub fn bt_net_hint_next_batch(net: *mut bt_net, next: *const bt_batch, ready_event: *mut c_void) -> c_int;
        pub fn bt_net_optimiser_step(net: *mut bt_net, stream: *mut c_void, adj_dev: *const f32, lr_dev: *const f32) -> c_int;
        pub fn bt_net_train_step(
            net: *mut bt_net, stream: *mut c_void, batch: *const bt_batch, loss_sum_dev: *mut f32, adj_dev: *const f32,
            lr_dev: *const f32, done_event: *mut c_void,
        ) -> c_int;
        pub fn bt_net_optimiser_step_event(
            net: *mut bt_net, stream: *mut c_void, adj_dev: *const f32, lr_dev: *const f32, done_event: *mut c_void,
        ) -> c_int;
        pub fn bt_net_forward(
            net: *mut bt_net, stream: *mut c_void, batch: *const bt_batch, out_dev: *mut f32, loss_sum_dev: *mut f32,
        ) -> c_int;
        pub fn bt_net_init_weights_host(
            desc: *const bt_net_desc, seed: u64, ft_effective_inputs: c_int, flat_host: *mut f32,
        ) -> c_int;
        pub fn bt_net_save_checkpoint(
            net: *mut bt_net, stream: *mut c_void, fmt: *const bt_saved_format, n_fmt: c_int, dir: *const c_char,
        ) -> c_int;
        pub fn bt_net_load_checkpoint(net: *mut bt_net, stream: *mut c_void, dir: *const c_char) -> c_int;

        // section 7: data-parallel exchange (RCCL bound at run time; bt_comm is opaque, passed as *mut c_void)
        pub fn bt_comm_unique_id(id128: *mut c_void) -> c_int;
        pub fn bt_comm_init_rank(id128: *const c_void, rank: c_int, world: c_int, comm: *mut *mut c_void) -> c_int;
        pub fn bt_comm_from_hooks(hooks: *const bt_comm_hooks, rank: c_int, world: c_int, comm: *mut *mut c_void) -> c_int;
        pub fn bt_comm_destroy(comm: *mut c_void) -> c_int;
        pub fn bt_comm_available() -> c_int;
        pub fn bt_comm_use_own_stream(comm: *mut c_void) -> c_int;
        pub fn bt_net_use_graphs(net: *mut bt_net, on: c_int) -> c_int;
        pub fn bt_allreduce_sum_f32(comm: *mut c_void, stream: *mut c_void, buf_dev: *mut f32, n: usize) -> c_int;
        pub fn bt_net_dp_plan(desc: *const bt_net_desc, world: c_int, rank: c_int, shard_min_floats: usize, plan: *mut bt_dp_plan) -> c_int;
        pub fn bt_net_dp_step(
            net: *mut bt_net, comm: *mut c_void, stream: *mut c_void, adj_dev: *const f32, lr_dev: *const f32, shard_min_floats: usize,
        ) -> c_int;
        pub fn bt_net_dp_gather_state(net: *mut bt_net, comm: *mut c_void, stream: *mut c_void, shard_min_floats: usize) -> c_int;

        pub fn bt_input_init(d: *mut bt_input_desc, kind: c_int, factorised: c_int, layout: *const c_int, layout_len: c_int) -> c_int;
        pub fn bt_input_num_inputs(d: *const bt_input_desc) -> c_int;
        pub fn bt_input_max_active(d: *const bt_input_desc) -> c_int;
        pub fn bt_map_batch(
            d: *const bt_input_desc, o: *const bt_map_opts, records: *const c_void, count: c_int, stm: *mut i32,
            nstm: *mut i32, buckets: *mut i32, targets: *mut f32, weights: *mut f32,
        ) -> c_int;
        pub fn bt_chessboard_from_str(line: *const c_char, record32: *mut c_void) -> c_int;

        pub fn bt_reader_open(paths: *const *const c_char, n_paths: c_int, start_position: u64, reader: *mut *mut bt_reader) -> c_int;
        pub fn bt_reader_open_viribinpack(
            paths: *const *const c_char, n_paths: c_int, buffer_size_mb: usize, threads: c_int, filter: bt_viri_filter,
            user: *mut c_void, shuffle_seed: u64, reader: *mut *mut bt_reader,
        ) -> c_int;
        pub fn bt_reader_open_sfbinpack(
            paths: *const *const c_char, n_paths: c_int, buffer_size_mb: usize, threads: c_int, filter: bt_sf_filter,
            user: *mut c_void, shuffle_seed: u64, reader: *mut *mut bt_reader,
        ) -> c_int;
        pub fn bt_viri_filter_default(filter: *mut bt_viri_builtin_filter);
        pub fn bt_viri_filter_unrestricted(filter: *mut bt_viri_builtin_filter);
        pub fn bt_reader_open_viribinpack_builtin(
            paths: *const *const c_char, n_paths: c_int, buffer_size_mb: usize, threads: c_int,
            filter: *const bt_viri_builtin_filter, shuffle_seed: u64, reader: *mut *mut bt_reader,
        ) -> c_int;
        pub fn bt_reader_open_montybinpack(
            paths: *const *const c_char, n_paths: c_int, buffer_size_mb: usize, threads: c_int, filter: bt_monty_filter,
            user: *mut c_void, shuffle_seed: u64, reader: *mut *mut bt_reader,
        ) -> c_int;
        pub fn bt_reader_next(reader: *mut bt_reader, dst_records: *mut c_void, count: c_int) -> c_int;
        pub fn bt_reader_close(reader: *mut bt_reader) -> c_int;

        pub fn bt_train_run(
            net: *mut bt_net, input: *const bt_input_desc, cfg: *const bt_train_config, paths: *const *const c_char,
            n_paths: c_int, on_batch: bt_batch_callback, on_superbatch: bt_superbatch_callback, user: *mut c_void,
        ) -> c_int;
    }
}

/// `G::Error: From<String>` (`crates/gpu/src/runtime/bindings.rs:67`): the library's thread-local message.
#[derive(Debug, Clone, PartialEq, Eq)]
pub struct Error(pub String);

impl From<String> for Error {
    fn from(s: String) -> Self {
        Error(s)
    }
}

fn check(rc: c_int) -> Result<(), Error> {
    if rc == 0 {
        Ok(())
    } else {
        Err(Error(unsafe { CStr::from_ptr(ffi::bt_last_error()) }.to_string_lossy().into_owned()))
    }
}

/// The lowered fwd + bwd + optimiser step of a `ValueTrainerBuilder` net: what `Function::new` + `prealloc` +
/// `Optimiser::new` set up in the reference (`crates/gpu/src/function.rs:73-196`, `crates/trainer/src/optimiser.rs:103-127`).
pub struct Net {
    raw: *mut ffi::bt_net,
    pub desc: ffi::bt_net_desc,
}

// one thread per device, as the reference's single main thread drives both streams (run.rs:156-250)
unsafe impl Send for Net {}

impl Net {
    pub fn new(desc: ffi::bt_net_desc, max_batch: usize) -> Result<Self, Error> {
        let mut raw = std::ptr::null_mut();
        check(unsafe { ffi::bt_net_create(&desc, max_batch as c_int, std::ptr::null_mut(), std::ptr::null_mut(), &mut raw) })?;
        Ok(Self { raw, desc })
    }

    /// `ModelWeights::new(&definition, seed)` with the reference's RNG stream (`crates/trainer/src/model/weights.rs:52-74`).
    pub fn init_weights(&mut self, seed: u64, ft_effective_inputs: usize) -> Result<(), Error> {
        let n = unsafe { ffi::bt_net_param_count(&self.desc) };
        let mut host = vec![0.0f32; n];
        check(unsafe { ffi::bt_net_init_weights_host(&self.desc, seed, ft_effective_inputs as c_int, host.as_mut_ptr()) })?;
        check(unsafe {
            ffi::bt_memcpy_h2d_async(ffi::bt_net_params(self.raw).cast(), host.as_ptr().cast(), n * 4, std::ptr::null_mut())
        })?;
        check(unsafe { ffi::bt_stream_sync(std::ptr::null_mut()) })
    }

    /// `Function::execute` of the lowered backward graph (`function.rs:198-280`): enqueue-only.
    ///
    /// # Safety
    /// `batch`'s pointers, `loss_sum_dev` and `stream` must be valid device pointers / a HIP stream of this device.
    pub unsafe fn forward_backward(&mut self, stream: *mut c_void, batch: &ffi::bt_batch, loss_sum_dev: *mut f32) -> Result<(), Error> {
        check(ffi::bt_net_forward_backward(self.raw, stream, batch, loss_sum_dev))
    }

    /// `Optimiser::update` (`optimiser.rs:162-194`) with `gradient_factor` / `learning_rate` in 1-element device buffers.
    ///
    /// # Safety
    /// `adj_dev` and `lr_dev` must be device pointers to one `f32` each.
    pub unsafe fn optimiser_step(&mut self, stream: *mut c_void, adj_dev: *const f32, lr_dev: *const f32) -> Result<(), Error> {
        check(ffi::bt_net_optimiser_step(self.raw, stream, adj_dev, lr_dev))
    }

    /// `Optimiser::set_params_for_weight` (`optimiser.rs:204-206`)
    pub fn set_params_for_weight(&mut self, id: &str, decay: f32, beta1: f32, beta2: f32, min_w: f32, max_w: f32) -> Result<(), Error> {
        let n = unsafe { ffi::bt_net_num_tensors(&self.desc) };
        for t in 0..n {
            let name = unsafe { CStr::from_ptr(ffi::bt_net_tensor_name(&self.desc, t)) };
            if name.to_bytes() == id.as_bytes() {
                return check(unsafe { ffi::bt_net_set_adamw(self.raw, t, decay, beta1, beta2, min_w, max_w) });
            }
        }
        Err(Error("No weight with ID found!".to_string()))
    }

    pub fn save_to_checkpoint(&mut self, fmt: &[ffi::bt_saved_format], dir: &str) -> Result<(), Error> {
        let dir = CString::new(dir).map_err(|e| Error(e.to_string()))?;
        check(unsafe { ffi::bt_net_save_checkpoint(self.raw, std::ptr::null_mut(), fmt.as_ptr(), fmt.len() as c_int, dir.as_ptr()) })
    }

    pub fn load_from_checkpoint(&mut self, dir: &str) -> Result<(), Error> {
        let dir = CString::new(dir).map_err(|e| Error(e.to_string()))?;
        check(unsafe { ffi::bt_net_load_checkpoint(self.raw, std::ptr::null_mut(), dir.as_ptr()) })
    }

    pub fn raw(&mut self) -> *mut ffi::bt_net {
        self.raw
    }
}

impl Drop for Net {
    fn drop(&mut self) {
        unsafe { ffi::bt_net_destroy(self.raw) };
    }
}

/// A record source the training loop pulls from: bulletformat files, FEN text (`"text:<path>"`), or a binpack
/// decoder with the reference's shuffle buffer.
pub struct Reader {
    raw: *mut ffi::bt_reader,
}

impl Reader {
    pub fn viribinpack(paths: &[&str], buffer_size_mb: usize, threads: usize) -> Result<Self, Error> {
        let c: Vec<CString> = paths.iter().map(|p| CString::new(*p).unwrap()).collect();
        let p: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::bt_reader_open_viribinpack(p.as_ptr(), p.len() as c_int, buffer_size_mb, threads as c_int, None, std::ptr::null_mut(), 0, &mut raw)
        })?;
        Ok(Self { raw })
    }

    /// `ViriBinpackLoader::new(path, mb, threads, Filter::default())`: the crate's built-in filter with its defaults
    pub fn viribinpack_builtin(paths: &[&str], buffer_size_mb: usize, threads: usize, filter: Option<ffi::bt_viri_builtin_filter>) -> Result<Self, Error> {
        let c: Vec<CString> = paths.iter().map(|p| CString::new(*p).unwrap()).collect();
        let p: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
        let f = filter.unwrap_or_else(|| {
            let mut f = std::mem::MaybeUninit::<ffi::bt_viri_builtin_filter>::zeroed();
            unsafe {
                ffi::bt_viri_filter_default(f.as_mut_ptr());
                f.assume_init()
            }
        });
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::bt_reader_open_viribinpack_builtin(p.as_ptr(), p.len() as c_int, buffer_size_mb, threads as c_int, &f, 0, &mut raw)
        })?;
        Ok(Self { raw })
    }

    /// `MontyBinpackLoader::new(path, mb, threads, filter)`
    pub fn montybinpack(
        paths: &[&str], buffer_size_mb: usize, threads: usize, filter: ffi::bt_monty_filter, user: *mut c_void,
    ) -> Result<Self, Error> {
        let c: Vec<CString> = paths.iter().map(|p| CString::new(*p).unwrap()).collect();
        let p: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::bt_reader_open_montybinpack(p.as_ptr(), p.len() as c_int, buffer_size_mb, threads as c_int, filter, user, 0, &mut raw)
        })?;
        Ok(Self { raw })
    }

    pub fn sfbinpack(
        paths: &[&str], buffer_size_mb: usize, threads: usize, filter: ffi::bt_sf_filter, user: *mut c_void,
    ) -> Result<Self, Error> {
        let c: Vec<CString> = paths.iter().map(|p| CString::new(*p).unwrap()).collect();
        let p: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
        let mut raw = std::ptr::null_mut();
        check(unsafe {
            ffi::bt_reader_open_sfbinpack(p.as_ptr(), p.len() as c_int, buffer_size_mb, threads as c_int, filter, user, 0, &mut raw)
        })?;
        Ok(Self { raw })
    }

    pub fn raw(&mut self) -> *mut ffi::bt_reader {
        self.raw
    }
}

impl Drop for Reader {
    fn drop(&mut self) {
        unsafe { ffi::bt_reader_close(self.raw) };
    }
}

/// `run::train` (`crates/trainer/src/run.rs:75-265`) for one device: returns the running loss of every superbatch.
pub fn train(net: &mut Net, input: &ffi::bt_input_desc, cfg: &ffi::bt_train_config, paths: &[&str]) -> Result<Vec<f32>, Error> {
    unsafe extern "C" fn on_superbatch(user: *mut c_void, _superbatch: c_int, running_loss: f32, _pps: f32) {
        (*(user as *mut Vec<f32>)).push(running_loss);
    }
    let c: Vec<CString> = paths.iter().map(|p| CString::new(*p).unwrap()).collect();
    let p: Vec<*const c_char> = c.iter().map(|s| s.as_ptr()).collect();
    let mut losses: Vec<f32> = Vec::new();
    check(unsafe {
        ffi::bt_train_run(
            net.raw, input, cfg, if p.is_empty() { std::ptr::null() } else { p.as_ptr() }, p.len() as c_int, None,
            Some(on_superbatch), (&mut losses as *mut Vec<f32>).cast(),
        )
    })?;
    Ok(losses)
}
