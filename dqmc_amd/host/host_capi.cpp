// placeholder translation unit until the C++ facade lands (next commit)
extern "C" int dqmc_host_abi_version(void) { return 1; }
