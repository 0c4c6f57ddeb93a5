// Links libqtts_hip.so.  The library is built from this repository with `make -C qwen3-tts-rs_amd/csrc`
// (hipcc --offload-arch=gfx950); QTTS_HIP_LIB_DIR names the directory that holds it.
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=QTTS_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    let dir = env::var("QTTS_HIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        // default: the in-tree build next to this crate
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("qwen3-tts-rs_amd")
    });
    assert!(
        dir.join("libqtts_hip.so").exists(),
        "libqtts_hip.so not found in {} (set QTTS_HIP_LIB_DIR or run `make -C qwen3-tts-rs_amd/csrc`)",
        dir.display()
    );
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=qtts_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    // the HIP runtime the library itself links against
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
}
