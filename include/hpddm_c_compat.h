/* hpddm_c_compat.h -- the reference's own C API (interface/HPDDM.h:66-118) exported by libhpddm_c_hip.so (K = double) and, compiled
 * with -DFORCE_COMPLEX like the reference's library, by libhpddm_c_hip_z.so (K = double _Complex, interface/HPDDM.h:34-50)
 * on top of libhpddm_hip.so, so that a C program written against HPDDM.h (examples/schwarz.c + examples/generate.c)
 * links and runs UNCHANGED with every subdomain factorised and solved on the MI355X.
 *
 * Mapping to the reference's execution model: one MPI rank owns one subdomain (HpddmSchwarzCreate is called once per
 * rank with that rank's matrix and neighbour lists, examples/schwarz.c:72); here every rank drives a one-subdomain
 * HpddmHipSchwarz whose halo and reductions travel through MPI (the transport callbacks of hpddm_hip.h filled with
 * MPI_Isend/Irecv and MPI_Allreduce -- the calls Subdomain::exchange and the Krylov methods make in the reference).
 * Ranks may share a GPU or own one each (HPDDM_HIP_DEVICE=<n>, default rank % device count).
 *
 * The whole of interface/HPDDM.h:66-118 is exported (the PETSc hook, #if HPDDM_PETSC, is not: no PETSc here).  A program includes
 * the reference's HPDDM.h as before -- this header only documents what the shim exports (same names, same argument
 * meaning) and lets the shim be compiled without the reference tree.
 */
#ifndef HPDDM_C_COMPAT_H_
#define HPDDM_C_COMPAT_H_
#include <mpi.h>
#include <stdbool.h>
#ifdef FORCE_COMPLEX /* one scalar type per library, as in the reference */
  #ifdef __cplusplus
    #include <complex>
typedef std::complex<double> HpddmK;
  #else
typedef double _Complex HpddmK;
  #endif
#else
typedef double HpddmK;
#endif
#ifdef __cplusplus
extern "C" {
#endif
typedef struct HpddmOption         HpddmOption;         /* interface/HPDDM.h:66-67 */
typedef struct HpddmMatrixCSR      HpddmMatrixCSR;      /* :80-81 */
typedef struct HpddmSubdomain      HpddmSubdomain;      /* :86-87 */
typedef struct HpddmPreconditioner HpddmPreconditioner; /* :92-93 */
typedef struct HpddmSchwarz        HpddmSchwarz;        /* :99-100 */

const HpddmOption *HpddmOptionGet(void);                                                      /* :68 */
int                HpddmOptionParse(const HpddmOption *, int, char **, bool);                 /* :69  -hpddm_* flags */
int                HpddmOptionParseString(const HpddmOption *, const char *);                 /* :70 */
int                HpddmOptionParseInt(const HpddmOption *, int, char **, char *, char *);    /* :71  application flags "name=<default>" */
int                HpddmOptionParseInts(const HpddmOption *, int, char **, int, char *[], char *[]); /* :72 */
int                HpddmOptionParseArgs(const HpddmOption *, int, char **, int, char *[], char *[]); /* :73  "name=(0|1)" */
bool               HpddmOptionSet(const HpddmOption *, const char *);                         /* :74 */
void               HpddmOptionRemove(const HpddmOption *, const char *);                      /* :75 */
double             HpddmOptionVal(const HpddmOption *, const char *);                         /* :76 */
double            *HpddmOptionAddr(const HpddmOption *, const char *);                        /* :77 */
double             HpddmOptionApp(const HpddmOption *, const char *);                         /* :78 */

HpddmMatrixCSR *HpddmMatrixCSRCreate(int n, int m, int nnz, HpddmK *a, int *ia, int *ja, bool sym, bool takeOwnership); /* :82 */
void            HpddmMatrixCSRDestroy(HpddmMatrixCSR *);                                      /* :83 */
void            HpddmCSRMM(HpddmMatrixCSR *, const HpddmK *, HpddmK *, int);                  /* :84 */

void HpddmSubdomainNumfact(HpddmSubdomain **, HpddmMatrixCSR *);                              /* :88 */
void HpddmSubdomainSolve(HpddmSubdomain *, const HpddmK *, HpddmK *, unsigned short);         /* :89 */
void HpddmSubdomainDestroy(HpddmSubdomain *);                                                 /* :90 */

void            HpddmInitializeCoarseOperator(HpddmPreconditioner *, unsigned short);         /* :94 */
void            HpddmSetVectors(HpddmPreconditioner *, HpddmK **);                            /* :95 */
void            HpddmDestroyVectors(HpddmPreconditioner *);                                   /* :96 */
const MPI_Comm *HpddmGetCommunicator(HpddmPreconditioner *);                                  /* :97 */

HpddmSchwarz        *HpddmSchwarzCreate(HpddmMatrixCSR *, int neighbors, int *list, int *sizes, int **connectivity); /* :101 */
void                 HpddmSchwarzInitialize(HpddmSchwarz *, double *d);                       /* :102 */
HpddmPreconditioner *HpddmSchwarzPreconditioner(HpddmSchwarz *);                              /* :103 */
void                 HpddmSchwarzMultiplicityScaling(HpddmSchwarz *, double *d);              /* :104 */
void                 HpddmSchwarzExchange(HpddmSchwarz *, HpddmK *, unsigned short);          /* :105 */
void                 HpddmSchwarzCallNumfact(HpddmSchwarz *);                                 /* :106 */
void                 HpddmSchwarzSolveGEVP(HpddmSchwarz *, HpddmMatrixCSR *neumann);          /* :107 */
void                 HpddmSchwarzBuildCoarseOperator(HpddmSchwarz *, MPI_Comm);               /* :108 */
void                 HpddmSchwarzComputeResidual(HpddmSchwarz *, const HpddmK *sol, const HpddmK *f, double *storage, unsigned short); /* :109 */
void                 HpddmSchwarzDestroy(HpddmSchwarz *);                                     /* :110 */

int HpddmSolve(HpddmSchwarz *, const HpddmK *b, HpddmK *sol, int mu, const MPI_Comm *);       /* :112 */
/* :113-115: the Krylov methods of -hpddm_krylov_method on an operator and a preconditioner given as callbacks on host vectors
 * (n x mu, column-major, n rows on this rank; inner products summed over *comm): interface/hpddm_c.cpp:41-53, 227-230.  Returns
 * the iteration count.  The basis and the recurrences live in HBM; every callback is one round trip over PCIe. */
typedef struct HpddmCustomOperator HpddmCustomOperator;
int HpddmCustomOperatorSolve(const HpddmCustomOperator *A, int n, int (*mv)(const HpddmCustomOperator *, const HpddmK *, HpddmK *, int), int (*precond)(const HpddmCustomOperator *, const HpddmK *, HpddmK *, int), const HpddmK *b, HpddmK *sol, int mu, const MPI_Comm *comm); /* both scalar types: libhpddm_c_hip.so (K = double) and libhpddm_c_hip_z.so (K = double _Complex, -DFORCE_COMPLEX) */

double nrm2(const int *, const HpddmK *, const int *);                                        /* :117 */
void   axpy(const int *, const HpddmK *, const HpddmK *, const int *, HpddmK *, const int *); /* :118 */
#ifdef __cplusplus
}
#endif
#endif
