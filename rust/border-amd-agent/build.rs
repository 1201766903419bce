//! Points the linker at libborder_amd.so (built by `python -m border_amd.build`, i.e. hipcc --offload-arch=gfx950).
//! BORDER_AMD_LIB_DIR = the directory that holds it (border_amd/ of the border_amd repository).
use std::env;

fn main() {
    println!("cargo:rerun-if-env-changed=BORDER_AMD_LIB_DIR");
    let dir = env::var("BORDER_AMD_LIB_DIR").unwrap_or_else(|_| "/opt/border_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=border_amd");
    // the library dlopens librccl lazily and needs libamdhip64 at load time; an rpath keeps `cargo run` self-contained
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
}
