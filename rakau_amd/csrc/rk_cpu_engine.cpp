// The CPU engine of include/rakau_amd/cpu_engine.hpp compiled for AVX-512 -> lib/librakau_amd_cpu512.so, a separate
// shared object (hidden visibility, loaded with RTLD_LOCAL by rk_cpu_engine_run() only on CPUs that have AVX-512), so that
// no function compiled with these flags can ever be picked by the linker for code that runs on other CPUs.
#include "rk_cpu_engine_impl.hpp"

#include <new>

#if !defined(RAKAU_AMD_CPU_AVX512)
#error "this translation unit must be compiled with -mavx512f -mavx512dq -mavx512vl -mavx2 -mfma"
#endif

extern "C" __attribute__((visibility("default"))) int rk_cpu_engine_entry(const rk_cpu_job *job) noexcept
{
    try {
        if (!job) {
            return RK_EINVAL;
        }
        rk_cpu::run_job(*job);
        return RK_OK;
    } catch (const std::invalid_argument &) {
        return RK_EINVAL;
    } catch (const std::bad_alloc &) {
        return RK_ENOMEM;
    } catch (...) {
        return RK_ERUNTIME;
    }
}
