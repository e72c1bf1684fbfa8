//! `P::MeshHandle` etc. are opaque `Copy + Eq + Hash` values (lib.rs:402-409); the C ABI identifies objects by u64.
//! `Interner` hands every distinct handle a dense id, starting at 1 (0 means "no texture" in `StMaterial`) and never
//! reuses one, so a handle that is removed and inserted again is a new object for the library, as in the reference.
use std::collections::HashMap;
use std::hash::Hash;

#[derive(Debug)]
pub(crate) struct Interner<H> {
    ids: HashMap<H, u64>,
    next: u64,
}

impl<H> Default for Interner<H> {
    fn default() -> Self {
        Self { ids: HashMap::new(), next: 1 }
    }
}

impl<H: Copy + Eq + Hash> Interner<H> {
    /// The id of `handle`, allocated on first use.
    pub fn id(&mut self, handle: H) -> u64 {
        if let Some(&id) = self.ids.get(&handle) {
            return id;
        }
        let id = self.next;
        self.next += 1;
        self.ids.insert(handle, id);
        id
    }

    /// The id of a handle that may not exist (removals of unknown handles are silent no-ops in the reference).
    pub fn get(&self, handle: H) -> Option<u64> {
        self.ids.get(&handle).copied()
    }

    pub fn forget(&mut self, handle: H) -> Option<u64> {
        self.ids.remove(&handle)
    }
}
