/*
 * TEST INFRASTRUCTURE ONLY -- loader for oracle/_ref/libspgpu_ref.so.
 *
 * oracle/_ref/libspgpu_ref.so is the reference's OWN host code (src/core/ell.c,
 * hell.c, hdia.cpp and core.c), compiled unmodified from /root/reference
 * against the CUDA headers this image ships (see oracle/Makefile).  core.c
 * references CUDA runtime functions (cudaStreamCreate, ...) for which the
 * image has no library, so the object can only be opened with LAZY symbol
 * binding; those functions are never called (only spgpuSizeOf and the format
 * converters are).  Python's ctypes always binds eagerly, hence this small
 * dlopen(RTLD_LAZY) shim, which hands function addresses to the tests.
 */
#include <dlfcn.h>
#include <stddef.h>

static void* g_ref;

/* Opens the reference build; returns 0 on success.  *error (optional) gets
 * dlerror()'s text. */
int orc_ref_open(const char* path, const char** error)
{
    if (!g_ref)
        g_ref = dlopen(path, RTLD_LAZY | RTLD_LOCAL);
    if (!g_ref && error)
        *error = dlerror();
    return g_ref ? 0 : 1;
}

/* Address of a reference function, NULL if absent. */
void* orc_ref_symbol(const char* name)
{
    return g_ref ? dlsym(g_ref, name) : NULL;
}
