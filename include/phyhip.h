/*
 * phyhip.h -- C ABI of the MI355X-native Felsenstein-pruning likelihood engine (libphyhip.so).
 *
 * This is the drop-in boundary of SURVEY.md section 8b: PhyML's only accelerator seam is the
 * `#ifdef BEAGLE` hook (src/lk.c:1300-1302, :585-587, :2327-2367; src/beagle_utils.c), whose foreign
 * calls are the BEAGLE C API.  Every entry point below names the call of that seam it replaces
 * (reference file:line) and keeps its argument order and meaning, so the glue a PhyML maintainer adds
 * is a rename.  The numerical semantics, however, are those of PhyML's own AVX path
 * (src/avx.c, src/lk.c) -- in particular PhyML's power-of-two rescaling rule (threshold 2^-256, factor
 * 2^256, one int per pattern and partial buffer: src/avx.c:460-513) instead of BEAGLE's scale
 * buffers, the 1e-100 floor / row renormalisation of transition matrices (src/models.c:293-298) and
 * the [l_min,l_max] clamp of rate-scaled branch lengths (src/lk.c:2296-2300) -- because the parity
 * target is the AVX path, not BEAGLE (the manual quotes 1e-4 disagreement for the latter).
 *
 * Plain C: ints, doubles, pointers and sizes only.  One host thread drives one instance (the
 * reference is single-threaded and non re-entrant).  All functions return
 * PHYHIP_SUCCESS (0) or a negative PHYHIP_ERROR_* code; phyhip_get_last_error() gives the text.
 * The glue prints it and calls Exit(), as src/beagle_utils.c:246-249 does.
 *
 * Data layouts (identical to the reference's host buffers, so uploads/downloads are memcpys):
 *   partials buffer   [pattern][category][state]  double     (t_edge::p_lk_left/p_lk_rght)
 *   tip partials      [pattern][state]            double 0/1 (t_edge::p_lk_tip_r)
 *   scale factors     [pattern]                   int        (t_edge::sum_scale_left/rght)
 *   transition matrix [category][from][to]        double     (t_edge::Pij_rr)
 *
 * Buffer index space (as in src/beagle_utils.c:108-113, lk.c:2221-2230): partial-buffer indices
 * 0..tipCount-1 are the tips, tipCount..partialsBufferCount-1 are internal edge sides.
 *
 * Execution model: phyhip_update_partials() only *queues* operations; the queue is flushed as ONE
 * kernel launch (the whole post-order traversal for every pattern tile) by the first call that needs
 * results: phyhip_calculate_edge_log_likelihoods*, phyhip_get_*, phyhip_update_eigen_lr,
 * phyhip_synchronize.  Callers need not know this; results are as if every call were synchronous.
 */
#ifndef PHYHIP_H
#define PHYHIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define PHYHIP_SUCCESS                         0
#define PHYHIP_ERROR_GENERAL                  (-1)
#define PHYHIP_ERROR_OUT_OF_MEMORY            (-2)
#define PHYHIP_ERROR_UNIDENTIFIED_EXCEPTION   (-3)
#define PHYHIP_ERROR_UNINITIALIZED_INSTANCE   (-4)
#define PHYHIP_ERROR_OUT_OF_RANGE             (-5)
#define PHYHIP_ERROR_NO_RESOURCE              (-6)   /* no gfx950 device visible */
#define PHYHIP_ERROR_NO_IMPLEMENTATION        (-7)
#define PHYHIP_ERROR_FLOATING_POINT           (-8)

#define PHYHIP_OP_NONE (-1)   /* BEAGLE_OP_NONE */

/* requirementFlags bit of phyhip_create_instance: build the sharded (multi-device) form even for a resource list of
   ONE device -- same code path, communicator of one rank (used by the single-GPU tests of that path). */
#define PHYHIP_FLAG_SHARDED (1L << 40)
#define PHYHIP_UNIQUE_ID_BYTES 128 /* sizeof(ncclUniqueId) */
/* requirementFlags bit: the categoryCount "categories" of the instance are the CLASSES of a mixture of class models
   (src/mixt.c; e.g. the four classes of LG4X): class c has its own eigen system and frequencies (eigenIndex / 
   stateFrequenciesIndex = c), its category rate is the class rate, every partials buffer carries one scale vector per
   class, and evaluations go through phyhip_calculate_class_mixture_*.  20 states x up to 4 classes, or 4 states x 1, 2 or 4
   classes; with a resource list the instance is sharded like any other (one all-reduce per mixture evaluation). */
#define PHYHIP_FLAG_CLASS_AXIS (1L << 41)
/* requirementFlags bit: the arithmetic of the reference's GENERIC partial-likelihood loop (Update_Partial_Lk_Generic,
   src/lk.c:1332-1587) as `phyml --cov` runs it on 4- or 20-state data (mod->use_m4mod, src/cl.c:753-757, src/lk.c:1303-1324):
   no all-ones shortcut -- a fully ambiguous subtree yields the rounded row sums of the transition matrices instead of exactly
   1.0 -- everything else as the default path (the same fused multiply-add chains; pinned by tests/golden/nucleic_cov_generic.phyg).
   Served by the plain, non-pipelined kernel: the door is there for parity, the reference itself runs it at half speed. */
#define PHYHIP_FLAG_GENERIC_LOOP (1L << 42)

/* BeagleOperation (src/beagle_utils.c:243).  The two scale-index fields are accepted and ignored:
   scale vectors are implicit, one per partials buffer, as in PhyML. */
typedef struct
{
  int destinationPartials;
  int destinationScaleWrite;
  int destinationScaleRead;
  int child1Partials;
  int child1TransitionMatrix;
  int child2Partials;
  int child2TransitionMatrix;
} phyhip_operation;

/* BeagleInstanceDetails (src/beagle_utils.c:84-95) */
typedef struct
{
  int  resourceNumber;     /* HIP device ordinal */
  char resourceName[64];   /* e.g. "AMD Instinct MI355X" */
  char implName[64];       /* "phyhip-gfx950" */
  long flags;
  int  computeUnits;
  long long globalMemBytes;
} phyhip_instance_details;

/* ---- instance lifetime ------------------------------------------------------------------ */

/* replaces beagleCreateInstance, src/beagle_utils.c:119-133.
   compactBufferCount, scaleBufferCount, preferenceFlags are accepted for signature compatibility;
   resourceList[0] (if given) is the HIP device ordinal, else env PHYHIP_DEVICE, else 0.
   MULTI-GPU: a resource list of G > 1 devices creates ONE sharded instance -- patterns are split into G
   contiguous ranges [g*P/G, (g+1)*P/G), one per listed device (a device may be listed more than once), everything
   per-pattern (tips, weights, invariant sites, partials, scale vectors, per-site outputs) is sliced / concatenated by
   the entry points below, the model and the transition matrices are replicated, and every scalar-returning evaluation
   (phyhip_calculate_edge_log_likelihoods, phyhip_calculate_eigen_lnl[_dlnl]) ends in ONE RCCL all-reduce over the
   communicators of ncclCommInitAll: {warning flag, lnL} for Lk (the sum of src/lk.c:856), {warning, lnL, dlnL} for dLk
   (src/lk.c:744-745).  The caller sees the same API and the same numbers; mixture evaluations (below) accept sharded class instances too.
   Returns the instance id (>= 0) or a negative error. */
int phyhip_create_instance(int tipCount, int partialsBufferCount, int compactBufferCount, int stateCount,
                           int patternCount, int eigenBufferCount, int matrixBufferCount, int categoryCount,
                           int scaleBufferCount, const int *resourceList, int resourceCount,
                           long preferenceFlags, long requirementFlags, phyhip_instance_details *returnInfo);

/* replaces beagleFinalizeInstance, src/beagle_utils.c:266 */
int phyhip_finalize_instance(int instance);

const char *phyhip_get_last_error(void);

/* ---- inputs ---------------------------------------------------------------------------------- */

/* replaces beagleSetTipPartials, src/beagle_utils.c:153.  inPartials is [pattern][state] with 0/1
   entries (src/lk.c:26-161); it is stored on the device as one byte per pattern (an index into a
   table of allowed-state masks).  Entries other than 0 and 1 are rejected (PHYHIP_ERROR_OUT_OF_RANGE). */
int phyhip_set_tip_partials(int instance, int tipIndex, const double *inPartials);

/* replaces beagleSetTipStates (BEAGLE compact form): state in [0,stateCount) or >= stateCount for a
   fully ambiguous character. */
int phyhip_set_tip_states(int instance, int tipIndex, const int *inStates);
/* One pattern of one tip rewritten in place: inPartials[stateCount] of 0.0 / 1.0, as a row of phyhip_set_tip_partials.
   Init_Partial_Lk_Tips_Double_One_Character (src/lk.c:2092), the tip rewrite of the leave-one-out cross-validation loops
   (src/mixt.c:4225-4258, src/cv.c:51-118).  Takes effect in stream order, without a host synchronisation for the single states and
   for "every state" (the hidden character of the leave-one-out loop: its table entry exists from instance creation); on
   20-state instances the FIRST use of any other state set adds an entry to the device's table of state sets, which costs one
   stream synchronisation.  A row with no state allowed is rejected (PHYHIP_ERROR_OUT_OF_RANGE).  Partial vectors that depend
   on the tip are the caller's to update, as in the reference. */
int phyhip_set_tip_partials_at_pattern(int instance, int tipIndex, int pattern, const double *inPartials);

/* replaces beagleSetPartials: upload an internal partials buffer ([pattern][category][state]). */
int phyhip_set_partials(int instance, int bufferIndex, const double *inPartials);

/* replaces beagleSetPatternWeights, src/beagle_utils.c:165 (re-callable: bootstrap re-weights,
   src/utilities.c:3945-3955).  Patterns with weight <= DBL_MIN are skipped in every sum. */
int phyhip_set_pattern_weights(int instance, const double *inPatternWeights);

/* replaces beagleSetCategoryRates / beagleSetCategoryWeights, src/beagle_utils.c:281-305 */
int phyhip_set_category_rates(int instance, const double *inCategoryRates);
int phyhip_set_category_weights(int instance, int categoryWeightsIndex, const double *inCategoryWeights);

/* replaces beagleSetStateFrequencies, src/beagle_utils.c:322-325 */
int phyhip_set_state_frequencies(int instance, int stateFrequenciesIndex, const double *inStateFrequencies);

/* replaces beagleSetEigenDecomposition, src/beagle_utils.c:383-386.  inEigenVectors = r_e_vect,
   inInverseEigenVectors = l_e_vect (row-major [state][state]); inEigenValues are the RAW eigenvalues
   of Q (mod->eigen->e_val, src/models.c:275 -- the log() in beagle_utils.c:376 is stale). */
int phyhip_set_eigen_decomposition(int instance, int eigenIndex, const double *inEigenVectors,
                                   const double *inInverseEigenVectors, const double *inEigenValues);

/* PhyML-specific knobs the BEAGLE API has no slot for.
   l_min,l_max: mod->l_min/l_max (src/init.c:711-714); br_len_mult: mod->br_len_mult->v;
   apply_lk_scaling: tree->apply_lk_scaling (src/utilities.h:993). */
int phyhip_set_phyml_options(int instance, double l_min, double l_max, double br_len_mult, int apply_lk_scaling);

/* +I model: mod->ras->invar, mod->ras->pinvar->v, data->invar[pattern] (src/lk.c:820-842,1226-1273).
   invar may be NULL when invar_model == 0. */
int phyhip_set_invariant_sites(int instance, int invar_model, double pinvar, const short *invar);

/* ---- transition matrices --------------------------------------------------------------------- */

/* replaces beagleUpdateTransitionMatrices, src/lk.c:2344.  For each i < count builds, on the device,
   all categories of matrix probabilityIndices[i] for edge length edgeLengths[i] (= b->l->v; the MAX(0,.)
   x rate x br_len_mult product and the clamp of src/lk.c:2296-2300 are applied here), with the
   floor / renormalise post-processing of src/models.c:293-298.  Derivative indices must be NULL. */
int phyhip_update_transition_matrices(int instance, int eigenIndex, const int *probabilityIndices,
                                      const int *firstDerivativeIndices, const int *secondDerivativeIndices,
                                      const double *edgeLengths, int count);

/* replaces beagleSetTransitionMatrix, src/lk.c:2360: upload b->Pij_rr ([category][from][to]) as computed
   by the host's own PMat() -- bit-exact by construction.  (phyhip_update_transition_matrices above builds the
   same doubles on the device: its exp() is the reference's libm's, phyml_amd/csrc/phyhip_exp.hpp.) */
int phyhip_set_transition_matrix(int instance, int matrixIndex, const double *inMatrix, double paddedValue);

/* replaces beagleGetTransitionMatrix, src/lk.c:2351 */
int phyhip_get_transition_matrix(int instance, int matrixIndex, double *outMatrix);

/* ---- the hot path ---------------------------------------------------------------------------- */

/* replaces beagleUpdatePartials, src/beagle_utils.c:245 (i.e. the body of Update_Partial_Lk,
   src/lk.c:1300-1302 -> src/avx.c:301-522).  Operations are executed in order; an operation may read
   the destination of an earlier one, but not its own (PHYHIP_ERROR_OUT_OF_RANGE: Update_Partial_Lk never
   updates in place).  cumulativeScaleIndex is ignored. */
int phyhip_update_partials(int instance, const phyhip_operation *operations, int operationCount,
                           int cumulativeScaleIndex);

/* replaces beagleCalculateEdgeLogLikelihoods, src/beagle_utils.c:344 (i.e. the site loop of Lk(),
   src/lk.c:590-645 + Lk_Core :767-861).  count must be 1.  parentBufferIndices[0] is b->p_lk_left,
   childBufferIndices[0] the right side (tip index if b->rght->tax), probabilityIndices[0] b->Pij_rr.
   Derivative outputs must be NULL (PhyML differentiates in the eigen basis: phyhip_calculate_eigen_*). */
int phyhip_calculate_edge_log_likelihoods(int instance, const int *parentBufferIndices,
                                          const int *childBufferIndices, const int *probabilityIndices,
                                          const int *firstDerivativeIndices, const int *secondDerivativeIndices,
                                          const int *categoryWeightsIndices, const int *stateFrequenciesIndices,
                                          const int *cumulativeScaleIndices, int count,
                                          double *outSumLogLikelihood, double *outSumFirstDerivative,
                                          double *outSumSecondDerivative);

/* Same evaluation, but the sum stays in device memory (deviceOut[0] = lnL) and the call does not synchronise
   (callers that run their own collective; the library's own multi-GPU forms are described at phyhip_create_instance
   and phyhip_comm_init_rank). */
int phyhip_calculate_edge_log_likelihoods_device(int instance, int parentBufferIndex, int childBufferIndex,
                                                 int probabilityIndex, double *deviceOut);

/* replaces beagleGetSiteLogLikelihoods, src/beagle_utils.c:355 (log-likelihood per pattern,
   tree->c_lnL_sorted) */
int phyhip_get_site_log_likelihoods(int instance, double *outLogLikelihoods);

/* The other per-pattern outputs of Lk_Core that host readers use (src/lk.c:855-857, 2791, 2801);
   any pointer may be NULL.  (Class-axis instances: fact_sum_scale holds categoryCount x patternCount ints, [class][pattern].) */
int phyhip_get_site_outputs(int instance, double *c_lnL_sorted, double *cur_site_lk,
                            double *unscaled_site_lk_cat, int *fact_sum_scale);

/* replaces beagleGetPartials, src/beagle_utils.c:252 (download hook for ancestral.c, cv.c, m4.c ...) */
int phyhip_get_partials(int instance, int bufferIndex, int scaleIndex, double *outPartials);

/* sum_scale_left/rght of a partials buffer ([pattern] ints) */
int phyhip_get_scale_factors(int instance, int bufferIndex, int *outScaleFactors);
int phyhip_set_scale_factors(int instance, int bufferIndex, const int *inScaleFactors);

/* tree->numerical_warning of the last edge evaluation (src/lk.c:847-851) */
int phyhip_get_numerical_warning(int instance, int *outWarning);

/* ---- mixtures of class models ---------------------------------------------- */

/* MIXT_Lk(b, mixt_tree), src/mixt.c:730-1160, for one partition element without +I: every class tree of the mixture
   (n_catg = 1, its own rate matrix / frequencies; src/mixt.c:2603-2640) is one instance whose single category rate is
   the class rate (mixt_tree->mod->ras->gamma_rr[parent_class_number], src/lk.c:2298).  Flushes the queued operations
   of every class instance, evaluates the given edge in each (the Lk_Core calls of src/mixt.c:997-1010) and combines
   the classes per pattern: 2^-sum rescaling, proba * r_mat_weight / rMatWeightSum * e_frq_weight / eFrqWeightSum /
   sumProbas (src/mixt.c:1048-1053), DBL_MIN floor, log, pattern weights of the FIRST instance.  All instances must sit
   on the same device with the same pattern count and one category; up to 64 classes (profile mixtures of the C10-C60 kind:
   one class tree per profile), PHYHIP_ERROR_OUT_OF_RANGE beyond.  The per-pattern log-likelihoods
   (mixt_tree->c_lnL_sorted) are left in the first instance (phyhip_get_site_log_likelihoods).
   GROUPS OF CLASSES: an entry of `instances` may also be an instance created with PHYHIP_FLAG_CLASS_AXIS (below): it stands for
   its C classes, in category order, evaluated by ONE traversal launch; parent / child / matrix indices stay per ENTRY, the
   per-class tables (classProba, rMatWeight, eFrqWeight) run over all classes in list order (entry 0's classes, then entry
   1's, ...; 64 in total).  A mixture of K classes is then ceil(K / 4) traversal launches per evaluation instead of K -- e.g.
   4 + 4 + 2 for ten classes, 4 + 1 for five; with three nucleotide classes 2 + 1 (the nucleotide class axis holds 1, 2 or 4).
   phyhip_calculate_mixture_eigen_lnl_dlnl takes the same lists. */
int phyhip_calculate_mixture_log_likelihood(const int *instances, int count, const int *parentBufferIndices,
                                            const int *childBufferIndices, const int *probabilityIndices,
                                            const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                            double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                            double *outSumLogLikelihood);

/* MIXT_dLk(&l, b, mixt_tree), src/mixt.c:2962-3340 (same restrictions): lnL and dlnL/dl of the mixture at length *l of
   the edge whose eigen-basis products every class instance holds (phyhip_update_eigen_lr on each class first, as
   MIXT_Update_Eigen_Lr does).  *l is clamped with the FIRST instance's [l_min,l_max] like src/lk.c:672-673; per class
   the length is scaled by its category rate and br_len_mult and clamped again (src/mixt.c:3056-3083).
   left/rightBufferIndices: the two sides of the class edges (their scale exponents enter the 2^-sum rescaling). */
int phyhip_calculate_mixture_eigen_lnl_dlnl(const int *instances, int count, const int *leftBufferIndices,
                                            const int *rightBufferIndices, double *l, const double *classProba,
                                            const double *rMatWeight, const double *eFrqWeight, double rMatWeightSum,
                                            double eFrqWeightSum, double sumProbas, double *outLnL, double *outDLnL);

/* +I mixtures (src/mixt.c:1079-1112, 3212-3275): the invariant class of the mixture is not a class tree of the device (PhyML
   skips it in every per-class loop); its share -- pi_inv[invar[pattern]] x pinvar mixed into the site likelihood, the
   derivative scaled by (1 - pinvar) -- enters the combination.  Set on the FIRST instance of the class list (or on the
   class-axis instance); invar_model == 0 switches it off.  The class instances themselves keep invar_model = 0. */
int phyhip_set_mixture_invariant_sites(int instance, int invar_model, double pinvar, const short *invar,
                                       const double *piInvariantClass);

/* The same two evaluations on ONE instance created with PHYHIP_FLAG_CLASS_AXIS (class c = category c): the queued
   partial updates of ALL classes and their edge evaluations are one traversal launch (one per class instance above),
   followed by the combination.  The class trees of PhyML's mixture share their topology and their call sequence
   (MIXT_Update_Partial_Lk / MIXT_Update_PMat_At_Given_Edge loop over them, src/mixt.c:1191-1250), so the glue queues an
   operation once -- when the first class tree reports it.  parent / child / matrix indices as for
   phyhip_calculate_edge_log_likelihoods; the per-pattern log-likelihoods are left for phyhip_get_site_log_likelihoods,
   the per-class likelihoods in unscaled_site_lk_cat ([pattern][class]) of phyhip_get_site_outputs. */
int phyhip_calculate_class_mixture_log_likelihood(int instance, int parentBufferIndex, int childBufferIndex, int probabilityIndex,
                                                  const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                                  double rMatWeightSum, double eFrqWeightSum, double sumProbas,
                                                  double *outSumLogLikelihood);
int phyhip_calculate_class_mixture_eigen_lnl_dlnl(int instance, int leftBufferIndex, int rightBufferIndex, double *l,
                                                  const double *classProba, const double *rMatWeight, const double *eFrqWeight,
                                                  double rMatWeightSum, double eFrqWeightSum, double sumProbas, double *outLnL,
                                                  double *outDLnL);
/* sum_scale vector of class classIndex of a partials buffer (class-axis instances; class 0 = phyhip_get_scale_factors) */
int phyhip_get_class_scale_factors(int instance, int bufferIndex, int classIndex, int *outScaleFactors);

/* ---- eigen-basis branch-length derivative (no BEAGLE counterpart in the seam) ---------------- */

/* Update_Eigen_Lr(b,tree), src/lk.c:1038-1114 / src/avx.c:21-105: fills the instance's dot_prod
   [pattern][category][state] from the two sides of an edge (either side may be a tip).  Flushes the queue first: on
   nucleotide instances up to 2 048 patterns the queued partial update(s) and the products are one launch -- mostly one
   command of the resident evaluator -- and the call returns when the products are in device memory; the products are
   always ordered before whatever the instance is asked next. */
int phyhip_update_eigen_lr(int instance, int leftBufferIndex, int rightBufferIndex);

/* dLk(&l,b,tree), src/lk.c:655-753: clamps *l to [l_min,l_max], returns lnL and dlnL/dl from dot_prod
   and the fact_sum_scale left by the preceding edge evaluation. */
int phyhip_calculate_eigen_lnl_dlnl(int instance, double *l, double *outLnL, double *outDLnL);

/* Lk(b,tree) with use_eigen_lr == YES, src/lk.c:592-603,625-629,866-950 */
int phyhip_calculate_eigen_lnl(int instance, double l, double *outLnL);

int phyhip_get_dot_prod(int instance, double *outDotProd);

/* ---- multi-GPU, one process per GPU (MPI-style hosts; PhyML's MPI build runs one process per rank) -------------- */

/* ncclGetUniqueId: rank 0 calls it and broadcasts the PHYHIP_UNIQUE_ID_BYTES bytes by whatever means the host has (MPI_Bcast). */
int phyhip_comm_get_unique_id(char *outId);
/* ncclCommInitRank on the instance's device and stream.  The instance holds this rank's contiguous pattern shard (the
   caller slices its inputs); from now on phyhip_calculate_edge_log_likelihoods and phyhip_calculate_eigen_lnl[_dlnl]
   return the sum over ALL ranks on every rank (one all-reduce per evaluation, as for the sharded instance). */
int phyhip_comm_init_rank(int instance, int nranks, int rank, const char *uniqueId);
/* number of ranks in the instance's communicator(s) (1: no communicator) */
int phyhip_comm_size(int instance, int *outRanks);
/* pattern range and device of shard `shard` of an instance; returns the number of shards (1 for a plain instance) */
int phyhip_get_shard_range(int instance, int shard, int *outDevice, int *outFirstPattern, int *outPatternCount);

/* ---- stream / timing plumbing ---------------------------------------------------------------- */

/* Run on the caller's HIP stream (e.g. torch's current stream) instead of the instance's own. */
int phyhip_set_stream(int instance, void *hipStream);
int phyhip_synchronize(int instance);

/* Kernel timing with HIP events on the instance's stream, around the traversal kernel only.
   enable != 0 starts (and resets) accumulation. */
int phyhip_profile(int instance, int enable);
int phyhip_profile_read(int instance, double *outTraversalMs, int *outLaunches, double *outSiteUpdates);
/* The collective path of the evaluations profiled since phyhip_profile(instance, 1) on a sharded instance or on a rank of
   phyhip_comm_init_rank: HIP events on the first shard's stream around {per-device local sum, ncclAllReduce, publish kernel} --
   the time from this rank's traversal kernel having ended to the reduced scalar being on its way to the host, which includes
   waiting for the slowest rank.  outRanks: the communicator's size (1: no communicator, nothing measured). */
int phyhip_profile_read_collective(int instance, double *outMs, int *outCount, int *outRanks);
/* Name of the traversal kernel of the last launch profiled since phyhip_profile(instance, 1), template arguments included, as
   rocprofv3 --kernel-trace shows it without the namespace (bench.py: `roofline.kernel`); empty before the first such launch.
   Sharded instances: the first shard's. */
int phyhip_profile_read_kernel(int instance, char *outName, int capacity);
/* Traffic model of the launches profiled since phyhip_profile(instance, 1): the bytes those traversal launches had to
   move if nothing but the kernel's own register forwarding saved any -- every result written once, every child read
   unless it is a tip (1 byte) or one of the two previous results.  The honest floor under the algorithmic byte count of
   SURVEY 8(d), which also charges the forwarded reads. */
int phyhip_profile_read_traffic(int instance, double *outReadBytes, double *outWriteBytes);
/* The eigen-basis kernels launched since phyhip_profile(instance, 1): Update_Eigen_Lr's kernel (src/lk.c:1038) and the
   dLk / eigen-basis Lk kernel (src/lk.c:655-753), milliseconds and launches of each (HIP events on the instance's stream). */
int phyhip_profile_read_eigen(int instance, double *outEigenLrMs, int *outEigenLrLaunches, double *outDlkMs, int *outDlkLaunches);

/* The resident evaluators (small nucleotide alignments, scalar wanted on the host): the launch-bound calls of a search --
   the chain of dLk calls of a branch-length optimisation (src/optimiz.c: Br_Len_Opt) and the short evaluations of SPR
   (src/spr.c:643-646: up to four matrices rebuilt, one or two partial updates, the edge likelihood) -- are served by
   workgroups that stay on the device and take each evaluation from a host-mapped command record instead of a kernel
   launch per call, whenever nothing else of the instance is known to be running on its stream.  Counters since the
   instance was created, out[0..3] for the dLk evaluator and out[4..7] for the short-evaluation one: evaluations served that
   way, launches of the resident workgroups, commands nobody answered (the evaluation was then launched the ordinary way),
   evaluations launched the ordinary way because work queued on the stream was not known to have finished.
   PHYHIP_RESIDENT=0 in the environment switches both off; PHYHIP_RESIDENT_IDLE_US (default 1000) is how long the workgroups
   wait for a command before they leave. */
int phyhip_get_resident_stats(int instance, long long out[8]);

/* The same four counters for the large-grid resident evaluator (phyml_amd/csrc/phyhip_big.hpp): nucleotide instances of more
   than 64 pattern tiles (~2 000 patterns) whose scalar-returning calls -- SPR candidates, Lk(b), Update_Eigen_Lr, dLk -- are
   served by one persistent workgroup per compute unit instead of a launch per call.  One instance per device at a time holds
   those workgroups; they leave when the instance launches anything else, or after PHYHIP_RESIDENT_IDLE_US without a command. */
int phyhip_get_big_resident_stats(int instance, long long out[4]);

/* Virtual buffers.  A tip x tip partial vector ("cherry": both children of the node are tips, src/avx.c:527-549 Exex) is two
   matrix columns and one product per pattern -- cheaper to recompute in registers than to write (C*S*8+4 bytes per pattern)
   and read back.  A traversal launch of at least `minOperations` operations therefore does not STORE such results when every
   reader of them sits later in the same launch: the defining operation is issued in front of each reader instead, its result
   forwarded in registers, and the buffer's memory stays stale ("virtual").  Whatever reads such a buffer later -- a queued
   operation, an evaluation edge, phyhip_update_eigen_lr, phyhip_get_partials / _scale_factors, a mixture evaluation -- first
   gets the defining operation queued again, storing: every value that leaves through this interface is the double the reference
   has in t_edge::p_lk_* at that point (tests/test_gpu_virtual.py).  A matrix or tip row the definition reads cannot change
   under it: before one changes, its old value is moved to a snapshot slot of the buffer (whole-tree batches of device-built
   matrices, uploaded matrices) or the dependants are stored first.  Only launches of the list form take part (at least three operations stay in the launch, whatever
   minOperations says).  minOperations = 0 switches the feature off (and stores what
   is virtual); the default is 16, so the short launches of a tree search never leave anything virtual -- and the first of
   them that reads a virtual buffer stores them all, in its own launch.  Not on class-axis or generic-loop instances.  Sharded
   instances: applied to every shard. */
int phyhip_set_virtual_buffers(int instance, int minOperations);

/* out[0] buffers virtual right now, out[1] stores skipped so far (operations that left their result virtual), out[2] non-storing
   re-issues in front of readers, out[3] storing re-issues (materialisations).  Sharded instances: the first shard's counters. */
int phyhip_get_virtual_stats(int instance, long long out[4]);

#ifdef __cplusplus
}
#endif
#endif /* PHYHIP_H */
