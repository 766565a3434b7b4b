// build.rs — builds libskinrt.so with hipcc (through renderer_amd/csrc/Makefile) and links it.
//
//   SKINRT_SRC      path to renderer_amd/csrc of this repository (default: ../../renderer_amd/csrc)
//   SKINRT_PREBUILT set to a directory that already holds libskinrt.so to skip the hipcc build
//   HIPCC           compiler (default /opt/rocm/bin/hipcc), forwarded to make
//
// hipcc cross-compiles gfx950 code objects without a GPU, so this works on a build machine; the
// library itself refuses to initialise without an MI355X (skinrt_init -> SKINRT_E_DEVICE): there is
// no CPU fallback behind these bindings.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let manifest = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap());
    println!("cargo:rerun-if-env-changed=SKINRT_SRC");
    println!("cargo:rerun-if-env-changed=SKINRT_PREBUILT");
    if let Ok(dir) = env::var("SKINRT_PREBUILT") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-lib=dylib=skinrt");
        return;
    }
    let src = env::var("SKINRT_SRC").map(PathBuf::from).unwrap_or_else(|_| manifest.join("../../renderer_amd/csrc"));
    let src = src.canonicalize().expect("SKINRT_SRC does not exist");
    let mut make = Command::new("make");
    make.arg("-C").arg(&src).arg("ARCH=gfx950");
    if let Ok(hipcc) = env::var("HIPCC") {
        make.arg(format!("HIPCC={hipcc}"));
    }
    let status = make.status().expect("cannot run make");
    assert!(status.success(), "hipcc build of libskinrt.so failed");
    let lib = src.join("../lib").canonicalize().expect("renderer_amd/lib missing after the build");
    println!("cargo:rustc-link-search=native={}", lib.display());
    println!("cargo:rustc-link-lib=dylib=skinrt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib.display());
    for f in ["api.hip", "kernels_pose.hip", "kernels_skin.hip", "kernels_geom.hip", "kernels_export.hip", "placement.hip",
              "rig_compile.cpp", "rig_compile.hpp", "device_types.hpp", "Makefile"] {
        println!("cargo:rerun-if-changed={}", src.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", src.join("../../include/skinrt.h").display());
}
