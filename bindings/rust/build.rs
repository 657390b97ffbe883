// Link against the library built by `make -C labrador_ldpc_amd/csrc` and the HIP runtime the process supplies
// (the library itself is linked -no-hip-rt).
//   LABRADOR_LDPC_HIP_LIB_DIR  directory holding liblabrador_ldpc_hip.so   (default: ../../labrador_ldpc_amd)
//   ROCM_PATH                  ROCm installation                           (default: /opt/rocm)
use std::env;
use std::path::PathBuf;

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    let lib_dir = env::var("LABRADOR_LDPC_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| manifest.join("../../labrador_ldpc_amd"));
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=labrador_ldpc_hip");
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    println!("cargo:rerun-if-env-changed=LABRADOR_LDPC_HIP_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
}
