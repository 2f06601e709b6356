// Links libjfhip.so.  JFHIP_LIB_DIR = the directory that holds it (jellyfish_amd/ after `make -C jellyfish_amd/csrc`).
fn main() {
    println!("cargo:rerun-if-env-changed=JFHIP_LIB_DIR");
    let dir = std::env::var("JFHIP_LIB_DIR").expect("set JFHIP_LIB_DIR to the directory holding libjfhip.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=jfhip");
    // the HIP runtime the library itself links
    let rocm = std::env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
