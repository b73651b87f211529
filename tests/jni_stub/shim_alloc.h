/* tests/jni_stub/shim_alloc.h -- TEST INFRASTRUCTURE ONLY: the JNI shim's allocations, counted by fake_jni.c */
#include <stddef.h>
#include <stdlib.h>   /* (before the -Dmalloc / -Dfree renames take effect on the shim's own code) */
void* t_malloc(size_t);
void t_free(void*);
