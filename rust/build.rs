// build.rs for Toyni with the MI355X backend (replaces the reference's nvcc-driving build.rs:27-118).
//
// UNVERIFIED BY rustc: the build image has no cargo/rustc.  Kept deliberately small.  What IS
// verified (tests/test_rust_dropin.py): the exact hipcc and ar commands below, run in a scratch crate layout
// (hip/ filled as INTEGRATION.md section 1 says), produce a libtoyni_hip.a that links into a host program with
// the libraries emitted at the bottom (-lamdhip64 -lstdc++) plus -lm -lpthread, which rustc supplies on every link line as platform
// libraries of std (the test, linking with bare gcc, passes them by hand).  It compiles toyni_hip.hip for gfx950 ONLY with hipcc, archives it, links amdhip64, and
// emits `has_hip` -- which, unlike the reference's unused `has_cuda`, src/ntt.rs
// really gates on, so `--features hip` on a box without ROCm still builds the CPU path.
use std::{env, path::PathBuf, process::Command};

fn hipcc() -> Option<PathBuf> {
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    let cand = PathBuf::from(&rocm).join("bin/hipcc");
    if cand.exists() {
        return Some(cand);
    }
    Command::new("hipcc").arg("--version").output().ok().filter(|o| o.status.success()).map(|_| PathBuf::from("hipcc"))
}

fn main() {
    println!("cargo:rerun-if-changed=hip/toyni_hip.hip");
    println!("cargo:rerun-if-changed=hip/ntt_kernels.hpp");
    println!("cargo:rerun-if-changed=hip/ntt_plan.hpp");
    println!("cargo:rerun-if-changed=hip/bb_field.hpp");
    println!("cargo:rerun-if-changed=hip/merkle_kernels.hpp");
    println!("cargo:rerun-if-changed=hip/multi_gpu.hpp");
    println!("cargo:rerun-if-changed=hip/prover_kernels.hpp");
    println!("cargo:rerun-if-changed=hip/toyni_hip.h");
    println!("cargo::rustc-check-cfg=cfg(has_hip)");
    if env::var_os("CARGO_FEATURE_HIP").is_none() {
        return;
    }
    let Some(hipcc) = hipcc() else {
        println!("cargo:warning=feature `hip` requested but hipcc was not found; building the CPU path only");
        return;
    };
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let obj = out.join("toyni_hip.o");
    let lib = out.join("libtoyni_hip.a");
    // gfx950 (MI355X) is the only target: no other --offload-arch, no CUDA path.
    let ok = Command::new(&hipcc)
        .args(["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", "hip", "-c", "hip/toyni_hip.hip", "-o"])
        .arg(&obj)
        .status()
        .map(|s| s.success())
        .unwrap_or(false);
    if !ok {
        println!("cargo:warning=hipcc failed; building the CPU path only");
        return;
    }
    let ok = Command::new("ar").arg("rcs").arg(&lib).arg(&obj).status().map(|s| s.success()).unwrap_or(false);
    if !ok {
        println!("cargo:warning=ar failed; building the CPU path only");
        return;
    }
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=static=toyni_hip");
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-lib=dylib=stdc++");
    println!("cargo:rustc-cfg=has_hip");
}
