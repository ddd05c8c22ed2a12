// comm.cpp — RCCL final gather (filled in with the multi-GPU milestone)
#include "lgmi_internal.h"
namespace lgmi { int set_error(int code, const char* msg); }
extern "C" int lgmi_comm_unique_id(void*) { return lgmi::set_error(LGMI_E_STATE, "comm not built yet"); }
extern "C" int lgmi_comm_init(lgmi_ctx*, const void*, int, int) { return lgmi::set_error(LGMI_E_STATE, "comm not built yet"); }
extern "C" int lgmi_comm_allgather_u64(lgmi_ctx*, uint64_t, uint64_t*) { return lgmi::set_error(LGMI_E_STATE, "comm not built yet"); }
extern "C" int lgmi_comm_gather_rows(lgmi_ctx*, const lgmi_dresult*, int, lgmi_result*) { return lgmi::set_error(LGMI_E_STATE, "comm not built yet"); }
extern "C" void lgmi_comm_destroy(lgmi_ctx*) {}
