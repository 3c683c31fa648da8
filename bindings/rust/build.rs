// Links libpqv_hip.so (built by `make -C pq-vector_amd/csrc`).  PQV_LIB_DIR = directory holding the library;
// at run time libamdhip64 (ROCm) must be on the loader path.
fn main() {
    let dir = std::env::var("PQV_LIB_DIR").unwrap_or_else(|_| "../../pq-vector_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=pqv_hip");
    println!("cargo:rerun-if-env-changed=PQV_LIB_DIR");
}
