// Builds the two shared libraries in-tree with the repository's Makefile (hipcc --offload-arch=gfx950 for the kernels,
// g++ for the host library) and tells cargo where they are.  Hand-written HIP for gfx950 only: no CUDA headers, no
// second code path, nothing to configure.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../ray-tracer_amd");
    let ok = Command::new("make")
        .arg("-C")
        .arg(&root)
        .arg("lib/librtamd.so")
        .arg("lib/librtamd_host.so")
        .status()
        .expect("make is needed to build librtamd.so")
        .success();
    assert!(ok, "the hipcc / g++ build of librtamd.so and librtamd_host.so failed");
    let lib = root.join("lib");
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=rtamd_host");
    println!("cargo:rustc-link-lib=dylib=rtamd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib.display());
    println!("cargo:rerun-if-changed={}", root.join("csrc").display());
    println!("cargo:rerun-if-changed={}", root.join("host").display());
    println!("cargo:rerun-if-changed=../include/rtamd.h");
    println!("cargo:rerun-if-changed=../include/rtamd_host.h");
}
