// Compiles the HIP sources for gfx950 into libptrt.so and links it.
//
// The flags are part of the numerical contract of the kernels (no FMA contraction, no fast-math, correctly
// rounded division and square root, f32 subnormals kept): they are the FLAGS line of path-tracer_amd/csrc/Makefile.
// PTRT_SRC points at that directory (default: three levels up from this crate, i.e. the repository layout);
// HIPCC overrides the compiler path.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let out = PathBuf::from(env::var("OUT_DIR").expect("OUT_DIR"));
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").expect("CARGO_MANIFEST_DIR"));
    let src = env::var("PTRT_SRC")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("../../path-tracer_amd/csrc"));
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".to_string());
    let sources = ["pt_api.hip", "pt_kernels.hip", "pt_group.hip", "pt_host.cpp", "pt_bvh_sah.cpp"];
    let headers = ["pt_kernels.h", "pt_layout.h", "pt_math.h", "../../include/ptrt.h"];
    for f in sources.iter().chain(headers.iter()) {
        println!("cargo:rerun-if-changed={}", src.join(f).display());
    }
    println!("cargo:rerun-if-env-changed=PTRT_SRC");
    println!("cargo:rerun-if-env-changed=HIPCC");
    let lib = out.join("libptrt.so");
    let status = Command::new(&hipcc)
        .args([
            "-DPT_TRACE_WAVES_PER_EU=5",
            "--offload-arch=gfx950",
            "-O3",
            "-std=c++17",
            "-fPIC",
            "-shared",
            "-ffp-contract=off",
            "-fno-fast-math",
            "-fhip-fp32-correctly-rounded-divide-sqrt",
            "-fno-gpu-flush-denormals-to-zero",
            "-mfma",
            "-fno-slp-vectorize",
        ])
        .args(sources.iter().map(|f| src.join(f)))
        .arg("-ldl")
        .arg("-o")
        .arg(&lib)
        .status()
        .unwrap_or_else(|e| panic!("could not run {hipcc}: {e}"));
    assert!(status.success(), "hipcc failed building {}", lib.display());
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=ptrt");
    // so that binaries find the library next to the build output without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", out.display());
}
