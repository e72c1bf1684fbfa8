//! Multi-GPU for a Rust host (BASELINE.json configs 4 and 5): one process per GPU, one `Engine` per process, the scene
//! replicated; the frame is cut into tiles (`st_dist_partition`: 2 ranks row bands, 4: 2 x 2, 8: 4 x 2), every rank renders
//! its tile + apron, and the ONE collective of the path — the gather of the composed tiles to rank 0 — runs inside the
//! library over RCCL (grouped ncclSend / ncclRecv on a communication stream the engine owns). The host only has to carry
//! rank 0's 128-byte id to the other processes (a pipe, MPI, a file: whatever launched them).
//!
//! ```ignore
//! let id = if rank == 0 { let id = dist::unique_id()?; share(&id); id } else { receive() };
//! // SAFETY: `engine` is declared before `node`, so `node` is dropped first (its Drop calls st_dist_shutdown on the engine's handle)
//! let mut node = unsafe { dist::Node::join(engine.raw_handle(), rank, world, &id)? };
//! let (owned, window) = node.partition(camera, 0 /* default grid */, 16 /* apron */)?;
//! loop {
//!     engine.tick(..); engine.render_into(camera, frames[k]);          // composes only `window`
//!     node.gather(camera, frames[k], if rank == 0 { full } else { null_mut() })?;   // returns at once; overlaps frame N+1
//!     k ^= 1;
//!     // every few frames: all ranks exchange their frame times and move the tile edges (cost-weighted grid)
//!     // let (grid2, owned, window) = node.rebalance(camera, w, h, &grid, &frame_ms_of_all_ranks, 16, 16)?;
//! }
//! ```
use crate::ffi;
use std::ffi::c_void;

#[derive(Debug)]
pub struct DistError(pub i32, pub String);

fn check(status: i32) -> Result<(), DistError> {
    if status == ffi::ST_OK {
        return Ok(());
    }
    let msg = unsafe { std::ffi::CStr::from_ptr(ffi::st_last_error()) }.to_string_lossy().into_owned();
    Err(DistError(status, msg))
}

/// rank 0: `ncclGetUniqueId` through the library
pub fn unique_id() -> Result<ffi::StDistUniqueId, DistError> {
    let mut id = ffi::StDistUniqueId { internal: [0; 128] };
    check(unsafe { ffi::st_dist_unique_id(&mut id) })?;
    Ok(id)
}

/// the tile `rank` owns (a pure function of the frame and the rank count: every process computes every tile)
pub fn tile(width: u32, height: u32, world: u32, cols: u32, rank: u32) -> Result<ffi::StDistRect, DistError> {
    let mut r = ffi::StDistRect::default();
    check(unsafe { ffi::st_dist_partition(width, height, world, cols, rank, &mut r) })?;
    Ok(r)
}

pub struct Node {
    engine: *mut ffi::StEngine,
    pub rank: i32,
    pub world: i32,
}

impl Node {
    /// joins the RCCL communicator on the engine's device
    ///
    /// # Safety
    /// `engine` must be a live engine handle and must outlive this `Node`: `Drop` calls `st_dist_shutdown` on it (declare the `Node` AFTER
    /// the engine wrapper, so that it is dropped first).
    pub unsafe fn join(engine: *mut ffi::StEngine, rank: i32, world: i32, id: &ffi::StDistUniqueId) -> Result<Self, DistError> {
        check(unsafe { ffi::st_dist_init(engine, rank, world, id) })?;
        Ok(Self { engine, rank, world })
    }
    /// this rank's tile (+ apron) becomes the camera's render window; returns (owned, window)
    pub fn partition(&mut self, camera: u64, cols: u32, apron: u32) -> Result<(ffi::StDistRect, ffi::StDistRect), DistError> {
        let (mut o, mut w) = (ffi::StDistRect::default(), ffi::StDistRect::default());
        check(unsafe { ffi::st_dist_set_partition(self.engine, camera, cols, apron, &mut o, &mut w) })?;
        Ok((o, w))
    }
    /// Cost-weighted tiles: from every rank's frame time (one f32 per rank, gathered by the host's own means, the same on every rank) the
    /// grid whose tiles would cost the same, edges moved by at most `max_step` pixels (<= the apron keeps every newly owned pixel's history
    /// warm); this rank's tile of it becomes the camera's window. `current`: st_dist_grid's equal split the first time, then what this returned.
    pub fn rebalance(&mut self, camera: u64, width: u32, height: u32, current: &ffi::StDistGrid, frame_ms: &[f32], apron: u32, max_step: u32)
        -> Result<(ffi::StDistGrid, ffi::StDistRect, ffi::StDistRect), DistError> {
        assert_eq!(frame_ms.len() as u32, current.cols * current.rows);
        let mut next = *current;
        check(unsafe { ffi::st_dist_grid_rebalance(width, height, current, frame_ms.as_ptr(), max_step, &mut next) })?;
        let (mut o, mut w) = (ffi::StDistRect::default(), ffi::StDistRect::default());
        check(unsafe { ffi::st_dist_set_grid(self.engine, camera, &next, apron, &mut o, &mut w) })?;
        Ok((next, o, w))
    }
    /// `frame`: the device buffer the frame was just composed into on `stream`; `full`: rank 0's assembled frame (may be `frame`)
    pub fn gather(&mut self, camera: u64, frame: *const c_void, full: *mut c_void, stream: ffi::hipStream_t) -> Result<(), DistError> {
        check(unsafe { ffi::st_dist_gather(self.engine, camera, frame, full, stream) })
    }
    /// blocks until the gather that read `frame` (null: every gather in flight) has landed
    pub fn wait(&mut self, camera: u64, frame: *const c_void) -> Result<(), DistError> {
        check(unsafe { ffi::st_dist_wait(self.engine, camera, frame, std::ptr::null_mut(), 1) })
    }
}

impl Drop for Node {
    fn drop(&mut self) {
        unsafe { ffi::st_dist_shutdown(self.engine) };
    }
}
