// Links libbullet_amd.so; BULLET_AMD_LIB_DIR points at the directory that holds it (bullet_amd/ in the source tree).
fn main() {
    let dir = std::env::var("BULLET_AMD_LIB_DIR").unwrap_or_else(|_| "../../../bullet_amd".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=bullet_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=BULLET_AMD_LIB_DIR");
}
