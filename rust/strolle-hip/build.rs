//! Links libstrolle_hip.so (built by `strolle_amd/csrc/Makefile`) and the HIP runtime it sits on.
//!   STROLLE_HIP_LIB_DIR  directory that holds libstrolle_hip.so   (default: ../../strolle_amd/csrc)
//!   ROCM_PATH            ROCm installation                         (default: /opt/rocm)
use std::env;
use std::path::PathBuf;

fn main() {
    let here = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("STROLLE_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| here.join("../../strolle_amd/csrc"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=strolle_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rerun-if-env-changed=STROLLE_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
