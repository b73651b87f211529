#!/bin/bash
# builds tests/jni_stub/fake_jni: the JNI shim compiled against the stub jni.h (its malloc/free counted) + the fake JNIEnv harness
set -e
here="$(cd "$(dirname "$0")" && pwd)"; root="$here/../.."
gcc -O1 -std=gnu11 -Wall -I"$here" -I"$root/include" -Dmalloc=t_malloc -Dfree=t_free -include "$here/shim_alloc.h" -c "$root/lz4-java_amd/jni/net_jpountz_lz4_LZ4HIPJNI.c" -o "$here/shim.o"
gcc -O1 -std=gnu11 -Wall -I"$here" -I"$root/include" -c "$here/fake_jni.c" -o "$here/fake_jni.o"
gcc "$here/fake_jni.o" "$here/shim.o" -L"$root/lz4-java_amd" -llz4hip -Wl,-rpath,"$root/lz4-java_amd" -Wl,-rpath,/opt/rocm/lib -o "$here/fake_jni"
