// Link against libmbls_hip.so (built by `python -m milagro_bls_amd.build`); MBLS_LIB_DIR points at the directory holding it.
fn main() {
    if let Ok(dir) = std::env::var("MBLS_LIB_DIR") {
        println!("cargo:rustc-link-search=native={}", dir);
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    }
    println!("cargo:rustc-link-lib=dylib=mbls_hip");
    println!("cargo:rerun-if-env-changed=MBLS_LIB_DIR");
}
