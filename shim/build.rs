// build.rs of the reference crate with the MI355X back end: links librtc_amd.so (built by `make -C raytracer_challenge_amd/csrc`).
fn main() {
    let dir = std::env::var("RTC_AMD_LIB_DIR").expect("set RTC_AMD_LIB_DIR to the directory that holds librtc_amd.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=rtc_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=RTC_AMD_LIB_DIR");
}
