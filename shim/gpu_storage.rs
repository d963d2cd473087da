//! `limitador/src/storage/gpu.rs` — `GpuStorage`, a `CounterStorage` backed by the MI355X counter
//! engine (`librl_engine.so`, C ABI in `include/rl_engine.h`).
//!
//! This file is the reference-side half of the drop-in boundary.  It is written against the reference
//! tree as of `/root/reference` (`limitador/src/storage/mod.rs:279-339`, `counter.rs`, `limit.rs`) and is
//! meant to be dropped in as `limitador/src/storage/gpu.rs` behind a cargo feature:
//!
//! ```toml
//! # limitador/Cargo.toml
//! [features]
//! gpu_storage = []
//! ```
//! ```rust,ignore
//! // limitador/src/storage/mod.rs, next to the other backends (mod.rs:10-24)
//! #[cfg(feature = "gpu_storage")]
//! pub mod gpu;
//! ```
//! ```rust,ignore
//! // limitador/build.rs
//! #[cfg(feature = "gpu_storage")]
//! {
//!     println!("cargo:rustc-link-search=native={}", std::env::var("RL_ENGINE_LIB_DIR").unwrap());
//!     println!("cargo:rustc-link-lib=dylib=rl_engine");
//! }
//! ```
//! and installed with `RateLimiter::new_with_storage(Box::new(GpuStorage::new(1 << 25, 1 << 16)?))`
//! (`lib.rs:330-334`).  It lives INSIDE the crate because `StorageErr` has private fields and no public
//! constructor (`storage/mod.rs:312-339`): like `disk/mod.rs:9-18` it builds the error in place.
//!
//! It could not be compiled in the build environment of this repository (no rustc / cargo there); the
//! same logic — identity interning, counter order, result mapping, micro-batching — is what
//! `limitador_amd/csrc/host/gpu_counter_storage.cpp` implements in C++ and what the GPU tests run the
//! reference's scenarios through (`tests/test_gpu_host_mirror.py`).
//!
//! What it does per call:
//!   * interns the identity of a `Limit` (namespace, seconds, conditions, variables — `limit.rs:177-214`;
//!     `Limit`'s own `Hash`/`Eq`) to a dense `u32` limit id, and the identity of a `Counter` (limit identity
//!     + `set_variables`, `counter.rs:123-138`) to an exact `u64` key: no hashing of identities, no
//!     fingerprint collisions;
//!   * keeps the request-side attributes the reference reads from `Counter.limit` (`max_value`, `seconds`;
//!     `counter.rs:64-66,76-78`) in the engine's limit table (`rl_limits_set`), following the caller when
//!     `max_value` changes (pinned by `lib.rs:760-790`);
//!   * orders a request's counters like `in_memory.rs:105,121` (counters of limits without variables first);
//!   * aggregates concurrent `check_and_update` callers into one device batch (`max_batch` requests or
//!     `max_delay`, whichever comes first): a batch is applied exactly like the reference applied to its
//!     requests one after another, with ONE clock value for the batch — read by the batch leader when the batch
//!     runs; a caller's own reading would be at most `max_delay` earlier.  (Per-request clocks make the engine
//!     commit the batch run by run of equal clocks: microsecond stamps turn a batch into single requests, and a
//!     run that fails leaves the runs before it applied — ADVICE r03);
//!   * resolves identities to keys INSIDE the leader's critical section, after the sweep that may have forgotten
//!     them, so that no queued request carries a key whose cell was just dropped;
//!   * stages every batch in arrays allocated once and pinned in place (`rl_host_register`): no allocation and no
//!     pageable copy per batch.

use crate::counter::Counter;
use crate::limit::Limit;
use crate::storage::{Authorization, CounterStorage, StorageErr};
use std::collections::{BTreeMap, HashMap, HashSet};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int};
use std::sync::{Arc, Condvar, Mutex};
use std::time::{Duration, Instant, SystemTime, UNIX_EPOCH};

// ---- include/rl_engine.h ------------------------------------------------------------------------
#[repr(C)]
struct RlEngine {
    _private: [u8; 0],
}

#[repr(C)]
struct RlConfig {
    device: i32,
    max_batch_hits: u32,
    capacity_cells: u64,
    max_limits: u32,
    flags: u32,
    hash_seed: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
struct RlLimitRow {
    max_value: u64,
    seconds: u64,
}

#[repr(C)]
#[derive(Clone, Copy)]
struct RlHit {
    key: u64,
    limit: u32,
    delta: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
struct RlCellRow {
    key: u64,
    limit: u32,
    reserved: u32,
    value: u64,
    expiry_us: u64,
}

const RL_OK: i32 = 0;
const RL_SIMPLE: u32 = 0x8000_0000;
const RL_CFG_AUTO_GROW: u32 = 1;

extern "C" {
    fn rl_engine_create(cfg: *const RlConfig, out: *mut *mut RlEngine) -> i32;
    fn rl_engine_destroy(e: *mut RlEngine);
    fn rl_last_error(e: *const RlEngine) -> *const c_char;
    fn rl_status_is_transient(status: i32) -> i32;
    fn rl_host_register(e: *mut RlEngine, ptr: *mut std::ffi::c_void, bytes: u64) -> i32;
    fn rl_host_unregister(e: *mut RlEngine, ptr: *mut std::ffi::c_void) -> i32;
    fn rl_limits_set(e: *mut RlEngine, first: u32, rows: *const RlLimitRow, n: u32) -> i32;
    fn rl_add_counter(e: *mut RlEngine, limit: u32, key: u64) -> i32;
    #[allow(clippy::too_many_arguments)]
    fn rl_check_and_update_batch_ex(
        e: *mut RlEngine,
        hits: *const RlHit,
        n_hits: u32,
        req_off: *const u32,
        n_req: u32,
        req_delta: *const u64,
        req_now_us: *const u64,
        now_us: u64,
        load_counters: i32,
        verdict: *mut u8,
        first_limited: *mut i32,
        remaining: *mut u64,
        expires_in_us: *mut u64,
    ) -> i32;
    fn rl_is_within_limits_batch_ex(
        e: *mut RlEngine,
        hits: *const RlHit,
        n_hits: u32,
        delta: *const u64,
        now_us: u64,
        within: *mut u8,
    ) -> i32;
    fn rl_update_counter_batch_ex(e: *mut RlEngine, hits: *const RlHit, n_hits: u32, delta: *const u64, now_us: u64) -> i32;
    fn rl_get_counters(e: *mut RlEngine, limit: u32, now_us: u64, out: *mut RlCellRow, cap: u64, n_out: *mut u64) -> i32;
    fn rl_delete_counters(e: *mut RlEngine, limit: u32) -> i32;
    fn rl_clear(e: *mut RlEngine) -> i32;
    fn rl_sweep_expired_rows(e: *mut RlEngine, now_us: u64, out: *mut RlCellRow, cap: u64, n_removed: *mut u64) -> i32;
}

// rl_status values this file looks at (include/rl_engine.h)
const RL_ERR_INVALID: i32 = -1;
const RL_ERR_MISSING_SIMPLE: i32 = -5;
const RL_ERR_KEY_LIMIT: i32 = -6;
const RL_ERR_BATCH_TOO_LARGE: i32 = -7;

// ---- errors ---------------------------------------------------------------------------------------
/// An `rl_status` with the engine's message; becomes a `StorageErr` (`storage/mod.rs:312-339`).
#[derive(Debug)]
pub struct GpuEngineError {
    pub status: i32,
    pub message: String,
}

impl std::fmt::Display for GpuEngineError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rl_engine status {}: {}", self.status, self.message)
    }
}

impl std::error::Error for GpuEngineError {}

impl From<GpuEngineError> for StorageErr {
    fn from(error: GpuEngineError) -> Self {
        // RL_ERR_DEVICE / RL_ERR_BUSY are worth a retry (rl_status_is_transient)
        let transient = unsafe { rl_status_is_transient(error.status) } != 0;
        Self {
            msg: format!("GPU counter engine error: {error}"),
            source: Some(Box::new(error)),
            transient,
        }
    }
}

// ---- identity interning -----------------------------------------------------------------------------
struct Interner {
    /// Limit identity -> limit id (`Limit`'s Hash/Eq ignore max_value, name and id: limit.rs:177-214)
    limit_ids: HashMap<Arc<Limit>, u32>,
    /// id -> the Limit as last seen (its max_value / name may change without a new identity) + its table row
    limits: Vec<(Arc<Limit>, RlLimitRow)>,
    /// Counter identity -> exact u64 key
    counter_keys: HashMap<(u32, BTreeMap<String, String>), u64>,
    /// key -> (limit id, the counter's variables): get_counters rebuilds `Counter`s from the engine's rows, and a
    /// swept / deleted cell's identity is forgotten through it
    by_key: HashMap<u64, (u32, BTreeMap<String, String>)>,
    key_seq: u64,
}

impl Interner {
    fn new() -> Self {
        Self {
            limit_ids: HashMap::new(),
            limits: Vec::new(),
            counter_keys: HashMap::new(),
            by_key: HashMap::new(),
            key_seq: 0,
        }
    }

    /// (limit id, row changed?) — a new identity gets the next dense id; a known one follows the caller's
    /// max_value (Storage::update_limit, storage/mod.rs:67-83) and name.
    fn limit_id(&mut self, limit: &Limit) -> (u32, bool) {
        if let Some(&id) = self.limit_ids.get(limit) {
            let entry = &mut self.limits[id as usize];
            let mut changed = false;
            if entry.1.max_value != limit.max_value() {
                entry.1.max_value = limit.max_value();
                changed = true;
            }
            if entry.0.max_value() != limit.max_value() || entry.0.name() != limit.name() {
                entry.0 = Arc::new(limit.clone());
            }
            return (id, changed);
        }
        let id = self.limits.len() as u32;
        let arc = Arc::new(limit.clone());
        self.limit_ids.insert(Arc::clone(&arc), id);
        self.limits.push((
            arc,
            RlLimitRow {
                max_value: limit.max_value(),
                seconds: limit.seconds(),
            },
        ));
        (id, true)
    }

    fn counter_key(&mut self, limit_id: u32, vars: &BTreeMap<String, String>) -> u64 {
        if let Some(&k) = self.counter_keys.get(&(limit_id, vars.clone())) {
            return k;
        }
        // exact and collision-free: a dense sequence number, scrambled by an odd multiplier (a bijection on
        // u64) so that table slots spread out; the engine's two reserved tags are skipped
        let key = loop {
            self.key_seq += 1;
            let k = self.key_seq.wrapping_mul(0x9E37_79B9_7F4A_7C15);
            if k < 0xFFFF_FFFF_FFFF_FFFE {
                break k;
            }
        };
        self.counter_keys.insert((limit_id, vars.clone()), key);
        self.by_key.insert(key, (limit_id, vars.clone()));
        key
    }

    /// The cell of `key` is gone (swept or deleted): forget the identity.  The key is never handed out again
    /// (`key_seq` only grows), so a later counter with the same identity simply gets a fresh key and a fresh cell —
    /// which is what the reference does when moka has evicted an entry (in_memory.rs:122-127).
    fn forget_key(&mut self, key: u64) {
        if let Some((limit_id, vars)) = self.by_key.remove(&key) {
            self.counter_keys.remove(&(limit_id, vars));
        }
    }

    /// Every qualified counter of `limit_id` is gone (delete_counters).  The simple counter's key stays: add_counter
    /// re-creates its cell under the same identity (in_memory.rs:241-257 keeps no trace either way).
    fn forget_limit(&mut self, limit_id: u32) {
        let gone: Vec<u64> = self
            .by_key
            .iter()
            .filter(|(_, (id, vars))| *id == limit_id && !vars.is_empty())
            .map(|(k, _)| *k)
            .collect();
        for k in gone {
            self.forget_key(k);
        }
    }
}

// ---- micro-batching of check_and_update -----------------------------------------------------------------
struct Pending {
    /// the request's counters in processing order (identities only: limit + set_variables); the batch LEADER turns
    /// them into wire records, under the interner lock it holds from the sweep to the end of the batch
    counters: Vec<Counter>,
    /// filled by the leader
    hits: Vec<RlHit>,
    /// position in the caller's Vec<Counter> of every hit (counters are reordered: simple first)
    order: Vec<usize>,
    delta: u64,
    load_counters: bool,
    result: Option<Result<Answer, GpuEngineError>>,
}

/// The arrays one device batch travels in: allocated once for the largest batch, pinned in place with
/// `rl_host_register` (the engine's copies are then DMA from / into these pages), reused by every leader.
struct Staging {
    hits: Vec<RlHit>,
    off: Vec<u32>,
    deltas: Vec<u64>,
    verdict: Vec<u8>,
    first: Vec<i32>,
    rem: Vec<u64>,
    exp: Vec<u64>,
    registered: Vec<*mut std::ffi::c_void>,
}

impl Staging {
    fn new(engine: *mut RlEngine, max_batch: usize, max_batch_hits: usize) -> Self {
        let mut st = Self {
            hits: vec![RlHit { key: 0, limit: 0, delta: 0 }; max_batch_hits],
            off: vec![0u32; max_batch + 1],
            deltas: vec![0u64; max_batch],
            verdict: vec![0u8; max_batch],
            first: vec![-1i32; max_batch],
            rem: vec![0u64; max_batch_hits],
            exp: vec![0u64; max_batch_hits],
            registered: Vec::new(),
        };
        // the Vecs are never resized, so the ranges stay where they are until drop (unregistered there); a range the
        // runtime refuses simply stays pageable
        let ranges: [(*mut std::ffi::c_void, usize); 7] = [
            (st.hits.as_mut_ptr() as *mut _, st.hits.len() * std::mem::size_of::<RlHit>()),
            (st.off.as_mut_ptr() as *mut _, st.off.len() * 4),
            (st.deltas.as_mut_ptr() as *mut _, st.deltas.len() * 8),
            (st.verdict.as_mut_ptr() as *mut _, st.verdict.len()),
            (st.first.as_mut_ptr() as *mut _, st.first.len() * 4),
            (st.rem.as_mut_ptr() as *mut _, st.rem.len() * 8),
            (st.exp.as_mut_ptr() as *mut _, st.exp.len() * 8),
        ];
        for (ptr, bytes) in ranges {
            if bytes > 0 && unsafe { rl_host_register(engine, ptr, bytes as u64) } == RL_OK {
                st.registered.push(ptr);
            }
        }
        st
    }

    fn release(&mut self, engine: *mut RlEngine) {
        for ptr in self.registered.drain(..) {
            unsafe { rl_host_unregister(engine, ptr) };
        }
    }
}

struct Answer {
    limited: bool,
    /// index into the caller's Vec<Counter> of the first limited counter
    first_limited: Option<usize>,
    /// per hit, in `order`: (remaining, expires_in_us) when load_counters
    loaded: Vec<(u64, u64)>,
}

struct BatchQueue {
    pending: Vec<(u64, Pending)>, // ticket, request
    done: HashMap<u64, Pending>,
    next_ticket: u64,
    leader_active: bool,
    first_arrival: Option<Instant>,
}

/// `CounterStorage` on the MI355X engine.
pub struct GpuStorage {
    engine: *mut RlEngine,
    interner: Mutex<Interner>,
    queue: Mutex<BatchQueue>,
    queue_cv: Condvar,
    /// serialises engine calls that are not batched (the engine has its own mutex too; this one keeps the
    /// interner and the engine's limit table in step)
    call: Mutex<()>,
    /// the batch leader's arrays (one leader at a time: `leader_active`)
    staging: Mutex<Staging>,
    max_batch: usize,
    /// hits one engine batch may carry (the engine's max_batch_hits)
    max_batch_hits: usize,
    max_delay: Duration,
    /// interned qualified counters beyond which a batch leader sweeps the expired cells first — the bound moka's
    /// `cache_size` gives the reference (in_memory.rs:205-212); 0 = never (the table grows, RL_CFG_AUTO_GROW)
    sweep_after: usize,
}

// The engine handle is only used under `call` / by the batch leader; the engine itself is thread-safe.
unsafe impl Send for GpuStorage {}
unsafe impl Sync for GpuStorage {}

impl GpuStorage {
    /// `capacity_cells`: table slots (replaces moka's `cache_size`, in_memory.rs:205-212; the table grows
    /// instead of evicting); `max_batch`: most requests one device batch aggregates.
    pub fn new(capacity_cells: u64, max_batch: usize) -> Result<Self, StorageErr> {
        Self::with_options(capacity_cells, max_batch, Duration::from_micros(200), 0)
    }

    pub fn with_options(capacity_cells: u64, max_batch: usize, max_delay: Duration, device: i32) -> Result<Self, StorageErr> {
        Self::with_all_options(capacity_cells, max_batch, max_delay, device, 4096, (capacity_cells / 2) as usize)
    }

    /// `max_limits`: rows of the engine's limit table (limit identities ever seen; ids are not recycled);
    /// `sweep_after`: see the field.
    pub fn with_all_options(
        capacity_cells: u64,
        max_batch: usize,
        max_delay: Duration,
        device: i32,
        max_limits: u32,
        sweep_after: usize,
    ) -> Result<Self, StorageErr> {
        // up to 32 counters per request, within what the engine's 24-bit hit index allows
        let max_batch_hits = max_batch.max(1).saturating_mul(32).min((1usize << 24) - 1);
        let cfg = RlConfig {
            device,
            max_batch_hits: max_batch_hits as u32,
            capacity_cells,
            max_limits,
            flags: RL_CFG_AUTO_GROW, // the reference's storage never refuses a counter
            hash_seed: 0x9E37_79B9_7F4A_7C15,
        };
        let mut engine: *mut RlEngine = std::ptr::null_mut();
        let rc = unsafe { rl_engine_create(&cfg, &mut engine) };
        if rc != RL_OK {
            return Err(GpuEngineError {
                status: rc,
                message: "rl_engine_create failed (no MI355X visible? the engine has no CPU path)".into(),
            }
            .into());
        }
        Ok(Self {
            engine,
            interner: Mutex::new(Interner::new()),
            queue: Mutex::new(BatchQueue {
                pending: Vec::new(),
                done: HashMap::new(),
                next_ticket: 0,
                leader_active: false,
                first_arrival: None,
            }),
            queue_cv: Condvar::new(),
            call: Mutex::new(()),
            staging: Mutex::new(Staging::new(engine, max_batch.max(1), max_batch_hits)),
            max_batch: max_batch.max(1),
            max_batch_hits,
            max_delay,
            sweep_after,
        })
    }

    /// Drop every qualified cell whose window has ended, and forget the interned identity of each (no reference
    /// analogue: it stands in for moka's capacity eviction, in_memory.rs:208-210, and never changes a decision of an
    /// unexpired counter).  Runs by itself once more than `sweep_after` qualified counters are interned.
    pub fn sweep_expired(&self) -> Result<u64, StorageErr> {
        let mut interner = self.interner.lock().unwrap();
        let _g = self.call.lock().unwrap();
        Ok(self.sweep_locked(&mut interner)?)
    }

    /// (both locks held)
    fn sweep_locked(&self, interner: &mut Interner) -> Result<u64, GpuEngineError> {
        let cap = interner.by_key.len();
        let mut rows = vec![RlCellRow::default(); cap];
        let mut removed = 0u64;
        self.check(unsafe { rl_sweep_expired_rows(self.engine, now_us(), rows.as_mut_ptr(), cap as u64, &mut removed) })?;
        for row in rows.iter().take((removed as usize).min(cap)) {
            interner.forget_key(row.key);
        }
        Ok(removed)
    }

    fn check(&self, rc: i32) -> Result<(), GpuEngineError> {
        if rc == RL_OK {
            return Ok(());
        }
        let message = unsafe { CStr::from_ptr(rl_last_error(self.engine)) }.to_string_lossy().into_owned();
        Err(GpuEngineError { status: rc, message })
    }

    /// Interns the counter and returns its wire record; uploads the limit row when it is new or changed.
    fn hit_of(&self, interner: &mut Interner, counter: &Counter, delta: u64) -> Result<RlHit, GpuEngineError> {
        let (id, row_changed) = interner.limit_id(counter.limit());
        if row_changed {
            let row = interner.limits[id as usize].1;
            self.check(unsafe { rl_limits_set(self.engine, id, &row, 1) })?;
        }
        let key = interner.counter_key(id, counter.set_variables());
        Ok(RlHit {
            key,
            limit: id | if counter.is_qualified() { 0 } else { RL_SIMPLE },
            // the 32-bit wire field; a delta beyond it travels in the call's u64 delta array
            delta: delta.min(u32::MAX as u64) as u32,
        })
    }

    /// One device batch for `batch` (all with the same load_counters flag), results stored in place.  A batch the
    /// engine REFUSES for what one request carries (a simple counter that was never add_counter'ed, a malformed hit:
    /// validation errors, decided before anything is applied — the batch has ONE clock, so it is one all-or-nothing
    /// engine command) is re-run request by request, so that only the offending caller gets the error — the reference
    /// fails that caller alone.
    fn run_batch(&self, st: &mut Staging, batch: &mut [(u64, Pending)]) {
        let rc = self.run_batch_once(st, batch);
        if batch.len() > 1 && matches!(rc, RL_ERR_INVALID | RL_ERR_MISSING_SIMPLE | RL_ERR_KEY_LIMIT) {
            for q in 0..batch.len() {
                self.run_batch_once(st, &mut batch[q..q + 1]);
            }
        }
    }

    /// -> the engine's status (every request of `batch` has its `result` set)
    fn run_batch_once(&self, st: &mut Staging, batch: &mut [(u64, Pending)]) -> i32 {
        let load = batch[0].1.load_counters;
        let n_req = batch.len();
        let mut n_hits = 0usize;
        let mut big_delta = false;
        st.off[0] = 0;
        for (q, (_, p)) in batch.iter().enumerate() {
            st.hits[n_hits..n_hits + p.hits.len()].copy_from_slice(&p.hits);
            n_hits += p.hits.len();
            st.off[q + 1] = n_hits as u32;
            st.deltas[q] = p.delta;
            big_delta |= p.delta > u32::MAX as u64;
        }
        let rc = {
            let _g = self.call.lock().unwrap();
            unsafe {
                rl_check_and_update_batch_ex(
                    self.engine,
                    st.hits.as_ptr(),
                    n_hits as u32,
                    st.off.as_ptr(),
                    n_req as u32,
                    if big_delta { st.deltas.as_ptr() } else { std::ptr::null() },
                    std::ptr::null(), // one clock for the whole batch
                    now_us(),         // read here, under the lock: clocks of consecutive batches never go backwards
                    load as c_int,
                    st.verdict.as_mut_ptr(),
                    st.first.as_mut_ptr(),
                    if load { st.rem.as_mut_ptr() } else { std::ptr::null_mut() },
                    if load { st.exp.as_mut_ptr() } else { std::ptr::null_mut() },
                )
            }
        };
        let outcome = self.check(rc);
        for (q, (_, p)) in batch.iter_mut().enumerate() {
            p.result = Some(match &outcome {
                Err(e) => Err(GpuEngineError {
                    status: e.status,
                    message: e.message.clone(),
                }),
                Ok(()) => {
                    let base = st.off[q] as usize;
                    Ok(Answer {
                        limited: st.verdict[q] != 0,
                        first_limited: if st.verdict[q] != 0 {
                            Some(p.order[st.first[q] as usize - base])
                        } else {
                            None
                        },
                        loaded: if load {
                            (0..p.hits.len()).map(|j| (st.rem[base + j], st.exp[base + j])).collect()
                        } else {
                            Vec::new()
                        },
                    })
                }
            });
        }
        rc
    }

    /// Enqueue one request and wait for its answer.  The first caller to find no leader becomes the leader:
    /// it waits until the batch is full or `max_delay` has passed since the first arrival, takes the whole
    /// queue (requests stay in arrival order), runs it as consecutive device batches that share the
    /// load_counters flag, and wakes everybody up.
    fn submit(&self, pending: Pending) -> Result<Answer, GpuEngineError> {
        let mut q = self.queue.lock().unwrap();
        let ticket = q.next_ticket;
        q.next_ticket += 1;
        if q.pending.is_empty() {
            q.first_arrival = Some(Instant::now());
        }
        q.pending.push((ticket, pending));
        self.queue_cv.notify_all();
        loop {
            if let Some(mut p) = q.done.remove(&ticket) {
                return p.result.take().expect("answered request");
            }
            if !q.leader_active && !q.pending.is_empty() {
                q.leader_active = true;
                // close the batch: full, or max_delay after its first request arrived
                loop {
                    let waited = q.first_arrival.map(|t| t.elapsed()).unwrap_or_default();
                    if q.pending.len() >= self.max_batch || waited >= self.max_delay {
                        break;
                    }
                    let (guard, _) = self.queue_cv.wait_timeout(q, self.max_delay - waited).unwrap();
                    q = guard;
                }
                let mut batch: Vec<(u64, Pending)> = std::mem::take(&mut q.pending);
                q.first_arrival = None;
                drop(q);
                // From here to the end of the batch the leader holds the interner: the sweep (the bound moka's cache_size
                // gives the reference: past it, the expired cells go first) forgets identities, delete_counters does too,
                // and both must happen-before or happen-after "identity -> key -> cell" of a whole batch, never in between
                // (a request carrying the key of a cell that was just dropped would re-create the cell under a key no
                // later request of the same counter resolves to: its hits would be lost to the limit).
                let mut interner = self.interner.lock().unwrap();
                if self.sweep_after > 0 && interner.by_key.len() > self.sweep_after {
                    let _g = self.call.lock().unwrap();
                    let _ = self.sweep_locked(&mut interner);
                }
                for (_, p) in batch.iter_mut() {
                    let mut hits = Vec::with_capacity(p.counters.len());
                    for c in &p.counters {
                        match self.hit_of(&mut interner, c, p.delta) {
                            Ok(h) => hits.push(h),
                            Err(e) => {
                                p.result = Some(Err(e)); // (the limit table refused the row: this caller alone fails)
                                break;
                            }
                        }
                    }
                    p.hits = hits;
                }
                let mut st = self.staging.lock().unwrap();
                // runs of equal load_counters (a property of the whole engine call), at most max_batch requests and
                // max_batch_hits counters each; a request with more counters than the engine's batch can carry is
                // answered here (the staging arrays are sized for max_batch_hits)
                let mut start = 0;
                while start < batch.len() {
                    if batch[start].1.result.is_some() {
                        start += 1;
                        continue;
                    }
                    if batch[start].1.hits.len() > self.max_batch_hits {
                        batch[start].1.result = Some(Err(GpuEngineError {
                            status: RL_ERR_BATCH_TOO_LARGE,
                            message: format!(
                                "a request of {} counters exceeds the engine's batch of {} hits",
                                batch[start].1.hits.len(),
                                self.max_batch_hits
                            ),
                        }));
                        start += 1;
                        continue;
                    }
                    let load = batch[start].1.load_counters;
                    let mut end = start + 1;
                    let mut n_hits = batch[start].1.hits.len();
                    while end < batch.len()
                        && end - start < self.max_batch
                        && batch[end].1.result.is_none()
                        && batch[end].1.load_counters == load
                        && n_hits + batch[end].1.hits.len() <= self.max_batch_hits
                    {
                        n_hits += batch[end].1.hits.len();
                        end += 1;
                    }
                    self.run_batch(&mut st, &mut batch[start..end]);
                    start = end;
                }
                drop(st);
                drop(interner);
                q = self.queue.lock().unwrap();
                for (t, p) in batch {
                    q.done.insert(t, p);
                }
                q.leader_active = false;
                self.queue_cv.notify_all();
                continue;
            }
            q = self.queue_cv.wait(q).unwrap();
        }
    }
}

impl Drop for GpuStorage {
    fn drop(&mut self) {
        if let Ok(mut st) = self.staging.lock() {
            st.release(self.engine);
        }
        unsafe { rl_engine_destroy(self.engine) };
    }
}

fn now_us() -> u64 {
    // SystemTime::now().duration_since(UNIX_EPOCH) in microseconds: atomic_expiring_value.rs:62-66
    SystemTime::now().duration_since(UNIX_EPOCH).map(|d| d.as_micros() as u64).unwrap_or(0)
}

impl CounterStorage for GpuStorage {
    // in_memory.rs:20-35
    fn is_within_limits(&self, counter: &Counter, delta: u64) -> Result<bool, StorageErr> {
        let hit = {
            let mut interner = self.interner.lock().unwrap();
            self.hit_of(&mut interner, counter, delta)?
        };
        let mut within = 0u8;
        let _g = self.call.lock().unwrap();
        self.check(unsafe {
            rl_is_within_limits_batch_ex(
                self.engine,
                &hit,
                1,
                if delta > u32::MAX as u64 { &delta } else { std::ptr::null() },
                now_us(),
                &mut within,
            )
        })?;
        Ok(within != 0)
    }

    // in_memory.rs:38-44: only limits without variables get a cell up front, (0, UNIX_EPOCH)
    fn add_counter(&self, limit: &Limit) -> Result<(), StorageErr> {
        let mut interner = self.interner.lock().unwrap();
        let (id, row_changed) = interner.limit_id(limit);
        if row_changed {
            let row = interner.limits[id as usize].1;
            self.check(unsafe { rl_limits_set(self.engine, id, &row, 1) })?;
        }
        if !limit.variables().is_empty() {
            return Ok(());
        }
        let key = interner.counter_key(id, &BTreeMap::new());
        let _g = self.call.lock().unwrap();
        self.check(unsafe { rl_add_counter(self.engine, id | RL_SIMPLE, key) })?;
        Ok(())
    }

    // in_memory.rs:47-69
    fn update_counter(&self, counter: &Counter, delta: u64) -> Result<(), StorageErr> {
        let hit = {
            let mut interner = self.interner.lock().unwrap();
            self.hit_of(&mut interner, counter, delta)?
        };
        let _g = self.call.lock().unwrap();
        self.check(unsafe {
            rl_update_counter_batch_ex(
                self.engine,
                &hit,
                1,
                if delta > u32::MAX as u64 { &delta } else { std::ptr::null() },
                now_us(),
            )
        })?;
        Ok(())
    }

    // in_memory.rs:72-156
    fn check_and_update(
        &self,
        counters: &mut Vec<Counter>,
        delta: u64,
        load_counters: bool,
    ) -> Result<Authorization, StorageErr> {
        if counters.is_empty() {
            return Ok(Authorization::Ok); // (the façade short-circuits before the storage: lib.rs:434-440)
        }
        // counters of limits without variables first, then the qualified ones, each in Vec order
        // (in_memory.rs:105,121)
        let mut order: Vec<usize> = (0..counters.len()).filter(|&i| !counters[i].is_qualified()).collect();
        order.extend((0..counters.len()).filter(|&i| counters[i].is_qualified()));
        // identities only: the batch leader resolves them to keys inside its critical section (see submit)
        let answer = self.submit(Pending {
            counters: order.iter().map(|&i| counters[i].clone()).collect(),
            hits: Vec::new(),
            order: order.clone(),
            delta,
            load_counters,
            result: None,
        })?;
        if load_counters {
            for (j, &i) in order.iter().enumerate() {
                let (remaining, expires_in_us) = answer.loaded[j];
                counters[i].set_remaining(remaining); // counter.rs:96-106
                counters[i].set_expires_in(Duration::from_micros(expires_in_us));
            }
        }
        Ok(if answer.limited {
            // the name of the first limited counter's limit (in_memory.rs:91-93,97-99)
            let i = answer.first_limited.expect("a limited request names its counter");
            Authorization::Limited(counters[i].limit().name().map(|n| n.to_owned()))
        } else {
            Authorization::Ok
        })
    }

    // in_memory.rs:159-187: every counter of the limits in the set whose ttl is > 0.  (The reference walks its two
    // maps and keeps what `limits` contains; asking the engine limit by limit returns the same set.)
    fn get_counters(&self, limits: &HashSet<Arc<Limit>>) -> Result<HashSet<Counter>, StorageErr> {
        let mut res = HashSet::new();
        let interner = self.interner.lock().unwrap();
        let _g = self.call.lock().unwrap();
        let now = now_us();
        for limit in limits {
            let Some(&id) = interner.limit_ids.get(limit.as_ref()) else {
                continue; // never seen: no counters
            };
            let wire = id | if limit.variables().is_empty() { RL_SIMPLE } else { 0 };
            let mut n = 0u64;
            self.check(unsafe { rl_get_counters(self.engine, wire, now, std::ptr::null_mut(), 0, &mut n) })?;
            let mut rows = vec![RlCellRow::default(); n as usize];
            if n > 0 {
                self.check(unsafe { rl_get_counters(self.engine, wire, now, rows.as_mut_ptr(), n, &mut n) })?;
                rows.truncate(n as usize);
            }
            for row in rows {
                // a row whose key the interner does not know (it cannot happen while identities are resolved and
                // forgotten under the leader's lock; a cell loaded behind the binding's back could) has no identity
                // to report: skipped, not rebuilt with empty variables
                let Some((_, known)) = interner.by_key.get(&row.key) else {
                    continue;
                };
                let vars: HashMap<String, String> = known.iter().map(|(k, v)| (k.clone(), v.clone())).collect();
                let mut counter = Counter::resolved_vars(Arc::clone(limit), vars).map_err(|e| GpuEngineError {
                    status: -1,
                    message: format!("cannot rebuild counter: {e}"),
                })?;
                // `max_value - value`, unchecked in the reference (in_memory.rs:166,178): wraps in release
                counter.set_remaining(limit.max_value().wrapping_sub(row.value));
                counter.set_expires_in(Duration::from_micros(row.expiry_us)); // ttl(now) > 0
                res.insert(counter);
            }
        }
        Ok(res)
    }

    // in_memory.rs:190-195,241-257
    fn delete_counters(&self, limits: &HashSet<Arc<Limit>>) -> Result<(), StorageErr> {
        let mut interner = self.interner.lock().unwrap();
        let _g = self.call.lock().unwrap();
        for limit in limits {
            let found = interner.limit_ids.get(limit.as_ref()).copied();
            if let Some(id) = found {
                let wire = id | if limit.variables().is_empty() { RL_SIMPLE } else { 0 };
                self.check(unsafe { rl_delete_counters(self.engine, wire) })?;
                interner.forget_limit(id); // the cells are gone: so are the interned identities of its counters
            }
        }
        Ok(())
    }

    // in_memory.rs:198-201: the reference clears only the limits without variables — so does the engine
    fn clear(&self) -> Result<(), StorageErr> {
        let _g = self.call.lock().unwrap();
        self.check(unsafe { rl_clear(self.engine) })?;
        Ok(())
    }
}
