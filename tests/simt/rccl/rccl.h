// tests/simt/rccl/rccl.h — TEST INFRASTRUCTURE: the handful of RCCL types and constants sl_comm.hip needs at compile time, for the SIMT-emulated
// build (the library resolves librccl at run time by dlopen; under the emulator that finds tests/simt's own stand-in or nothing).
// Values = the stable NCCL ABI, the same ones sl_comm.hip checks with static_assert.
#pragma once
#include <stddef.h>
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5,
               ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclSum = 0 } ncclRedOp_t;
typedef enum { ncclDouble = 8 } ncclDataType_t;
