fn main() {
    // directory holding libfrieda_hip.so (…/frieda_amd/lib after `python -c "import __graft_entry__ as g; g.build()"`)
    let dir = std::env::var("FRIEDA_HIP_LIB_DIR").expect("set FRIEDA_HIP_LIB_DIR to the directory of libfrieda_hip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=frieda_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=FRIEDA_HIP_LIB_DIR");
}
