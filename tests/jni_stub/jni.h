/* tests/jni_stub/jni.h -- TEST INFRASTRUCTURE ONLY: the handful of JNI declarations the shim uses, so
 * `gcc -fsyntax-only` can type-check the JNI shim sources in an image without a JDK.  Never shipped. */
#ifndef JNI_STUB_H
#define JNI_STUB_H
#include <stdint.h>
#define JNIEXPORT __attribute__((visibility("default")))
#define JNICALL
#define JNI_ABORT 2
typedef int32_t jint; typedef int32_t jsize; typedef int64_t jlong; typedef int8_t jbyte; typedef uint8_t jboolean;
typedef void* jobject; typedef jobject jclass; typedef jobject jstring; typedef jobject jarray;
typedef jarray jbyteArray; typedef jarray jintArray; typedef jarray jlongArray;
struct JNINativeInterface_;
typedef const struct JNINativeInterface_* JNIEnv;
struct JNINativeInterface_ {
  jclass (*FindClass)(JNIEnv*, const char*);
  jint (*ThrowNew)(JNIEnv*, jclass, const char*);
  jobject (*NewGlobalRef)(JNIEnv*, jobject);
  void* (*GetPrimitiveArrayCritical)(JNIEnv*, jarray, jboolean*);
  void (*ReleasePrimitiveArrayCritical)(JNIEnv*, jarray, void*, jint);
  void* (*GetDirectBufferAddress)(JNIEnv*, jobject);
  jstring (*NewStringUTF)(JNIEnv*, const char*);
  jlong* (*GetLongArrayElements)(JNIEnv*, jlongArray, jboolean*);
  jint* (*GetIntArrayElements)(JNIEnv*, jintArray, jboolean*);
  void (*ReleaseLongArrayElements)(JNIEnv*, jlongArray, jlong*, jint);
  void (*ReleaseIntArrayElements)(JNIEnv*, jintArray, jint*, jint);
  jint (*GetArrayLength)(JNIEnv*, jarray);
};
#endif
