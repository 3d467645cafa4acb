// Link against the in-tree HIP library.  LEANMULTISIG_HIP_DIR = directory holding libleanmultisig_hip.so.
fn main() {
    let dir = std::env::var("LEANMULTISIG_HIP_DIR").unwrap_or_else(|_| "../leanmultisig_amd".into());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=leanmultisig_hip");
    println!("cargo:rerun-if-env-changed=LEANMULTISIG_HIP_DIR");
}
