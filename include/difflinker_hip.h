/*
 * difflinker_hip.h — C ABI of the MI355X (gfx950) implementation of DiffLinker's EGNN
 * denoising-diffusion sampling hot path.
 *
 * The reference (igashov/DiffLinker) is pure Python/PyTorch and has no FFI of its own; the
 * entry points below are what a native replacement of its hot path binds, one per reference
 * interface (file:line into the reference tree):
 *
 *   dl_model_create / dl_model_destroy   <- Dynamics.__init__ + load_state_dict
 *                                           src/egnn.py:324-372, src/lightning.py:81-100
 *   dl_egnn_forward_fc                   <- Dynamics.forward (FC graph)       src/egnn.py:374-447
 *                                           (EGNN.forward :218-238, EquivariantBlock :157-178,
 *                                            GCL :45-80, EquivariantUpdate :101-125,
 *                                            coord2diff :295-301, unsorted_segment_sum :304-320)
 *   dl_egnn_forward_pocket               <- DynamicsWithPockets.forward       src/egnn.py:470-552
 *                                           (+ get_dist_edges / get_dist_edges_4A :554-596)
 *   dl_sampler_step                      <- EDM.sample_p_zs_given_zt_only_linker, the part after
 *                                           the denoiser call                   src/edm.py:198-208
 *   dl_sample_chain_fc                   <- EDM.sample_chain                  src/edm.py:126-176
 *                                           (+ :178-208 reverse step, :210-242 final decode,
 *                                            :328-361 noise / (un)normalisation)
 *   dl_workspace_bytes                   <- (no reference counterpart) scratch the fully-connected entry points need
 *   dl_egnn_forward_fc_team, dl_team_max, dl_team_max_atoms
 *                                        <- the same Dynamics.forward / EDM.sample_chain with several compute
 *                                           units per molecule (batches smaller than the chip; no reference
 *                                           counterpart: a launch-geometry knob, results agree to fp32 rounding)
 *   dl_size_model_create / dl_size_gnn_forward
 *                                        <- SizeGNN (src/linker_size.py:45-91) as driven by
 *                                           SizeClassifier.forward at inference
 *                                           (src/linker_size_lightning.py:83-110): the `sample_fn`
 *                                           of generate.py:86-99 that runs once before a chain
 *
 * Conventions
 *   - every pointer marked "device" is a HIP device pointer owned by the caller (PyTorch-ROCm
 *     tensors' data_ptr()); `stream` is a hipStream_t passed as void* (0 = default stream); calls are
 *     asynchronous on that stream.
 *   - the library allocates NOTHING after dl_model_create (which uploads the packed weights): every compute entry
 *     point takes its scratch memory from the caller - `workspace` / `workspace_bytes`, sized by the matching
 *     dl_*_workspace_bytes query, 16-byte aligned, contents irrelevant on entry, free to reuse once the launch has
 *     completed on `stream`.  Two launches that run concurrently need two workspaces; a dl_model itself is
 *     immutable after creation and may be shared by any number of streams.
 *   - all floating point is fp32; masks are int8 (node_mask, edge_mask) or fp32 (fragment /
 *     linker masks, context) exactly as the reference's collate produces them.
 *   - return value: 0 on success, a negative dl_status otherwise; dl_error_string() names it.
 *     There is NO CPU fallback: on a machine without a gfx950 device the compute entry points
 *     return DL_ERR_NO_DEVICE / a HIP error.
 */
#ifndef DIFFLINKER_HIP_H
#define DIFFLINKER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DL_ABI_VERSION 7

typedef enum dl_status {
    DL_OK = 0,
    DL_ERR_BAD_ARG = -1,        /* null pointer / inconsistent sizes                       */
    DL_ERR_UNSUPPORTED = -2,    /* hyper-parameter outside the HIP path (see dl_config)    */
    DL_ERR_TOO_MANY_ATOMS = -3, /* a molecule has more real atoms than dl_max_atoms()      */
    DL_ERR_HIP = -4,            /* a HIP runtime call failed (see dl_last_hip_error())     */
    DL_ERR_NO_DEVICE = -5,      /* no gfx950 device visible                                */
    DL_ERR_ALLOC = -6
} dl_status;

/* Arithmetic of the 128-wide contractions (edge/node/coordinate MLP layers).  Everything else
 * (distances, SiLU, masks, aggregation, sampler algebra) is fp32 in both modes.
 *   DL_PRECISION_FP32   v_mfma_f32_32x32x2_f32: exact fp32 FMA chains (runs at the fp32 vector rate)
 *   DL_PRECISION_F16X3  each fp32 operand is scaled by a power of two into the fp16 range and split into
 *                       fp16 hi+lo (~21 significant bits); a*w = hi*hi'+hi*lo'+lo*hi' on
 *                       v_mfma_f32_32x32x16_f16 with fp32 accumulation, rescaled exactly; ~1e-6 relative
 *                       per product (fp32-class), 2x faster than the fp32 MFMA on this workload
 *   DL_PRECISION_F16X2  (round 4, opt-in) F16X3 everywhere except the second layer of the GCL edge model (src/egnn.py:19-30,45-59),
 *                       where the first layer's activation enters as ONE fp16 rounded to nearest: a_rn*(hi'+lo'), 64 instead of
 *                       96 MFMAs per 32 pairs and no lo split (+9..11 % molecules/s).  Measured against the fp32 oracle: node
 *                       features 3e-6..9e-6 rel-L2 per forward (F16X3: 2e-7..5e-7), velocities and sampled coordinates
 *                       unchanged (<= 1e-6; the coordinate model stays F16X3) - inside the 1e-4 bar, outside fp32 class       */
typedef enum dl_precision { DL_PRECISION_FP32 = 0, DL_PRECISION_F16X3 = 1, DL_PRECISION_F16X2 = 2 } dl_precision;

/* Dynamics.__init__ hyper-parameters (src/egnn.py:324-329).  The HIP path implements model='egnn_dynamics' with SiLU on 128-wide
 * kernels: the released-config surface plus the optional attention, tanh, aggregation_method='mean' (every kernel family),
 * sin_embedding (HBM-resident kernels), 1..4 GCLs per block and an optional time feature (ABI v7).  A narrower network
 * (hidden_nf < 128; the reference's default is 64) is handed over ZERO-PADDED to 128 - the extra hidden features get zero weights
 * in and out and zero biases (SiLU(0) = 0): exactly the same function; difflinker_amd/egnn.py: pad_to_kernel_width does it. */
typedef struct dl_config {
    int32_t n_dims;               /* 3                                              */
    int32_t in_node_nf;           /* atom-type channels nf (8 ZINC, 9 GEOM/pockets) */
    int32_t context_node_nf;      /* 1..4                                           */
    int32_t hidden_nf;            /* must be 128 (narrower networks: zero-padded, above) */
    int32_t n_layers;             /* EquivariantBlocks (6 GEOM, 8 ZINC)             */
    int32_t inv_sublayers;        /* GCLs per EquivariantBlock, 1..4 (2 in every released configuration) */
    int32_t condition_time;       /* 1: the node inputs are [h, t, context] (released configurations); 0: [h, context] */
    float norm_constant;          /* 1e-6 in the released configs                   */
    float normalization_factor;   /* 100                                            */
    int32_t precision;            /* dl_precision: arithmetic of the 128-wide GEMMs */
    /* optional hyper-parameters of the reference no released configuration uses (every entry point carries them; round 3:
     * dl_egnn_forward_pocket and dl_egnn_forward_fc_large too): */
    int32_t attention;            /* GCL edge attention: m_ij *= sigmoid(w_att . m_ij + b_att)   src/egnn.py:42-43,52-54  */
    int32_t tanh;                 /* coordinate head: cdiff * tanh(s) * coords_range             src/egnn.py:104-105      */
    float coords_range;           /* 15 for Dynamics (EGNN hands its undivided default to the blocks, src/egnn.py:183,213) */
    int32_t aggregation_mean;     /* 0: sum / normalization_factor; 1: / number of edges of the row, masked ones included
                                   * (= the padded width N on the fully-connected graph, the atom's degree on a radius
                                   * graph)                                                        src/egnn.py:315-319      */
    int32_t sin_embedding;        /* 1: 24 sinusoidal edge attributes (src/egnn.py:281-292); the weights' edge-MLP input rows are
                                   * then [128][280].  HBM-resident entry points only (dl_egnn_forward_fc_large,
                                   * dl_egnn_forward_pocket); the LDS-resident ones answer DL_ERR_UNSUPPORTED               */
} dl_config;

typedef struct dl_model dl_model; /* opaque: packed, pre-scaled weights resident in HBM */

/* Number of weight tensors dl_model_create expects: 4 + n_layers * (inv_sublayers*(8 + 2*attention) + 5); with attention every GCL
 * appends att_mlp.0.weight [1,128] and att_mlp.0.bias [1] after its node_mlp tensors. */
int32_t dl_model_num_tensors(const dl_config* cfg);

/* Pack the reference's nn.Linear tensors ([out,in] row-major fp32, HOST pointers) into the
 * kernel layout and upload them.  `weights` lists the tensors in the reference state_dict
 * order of the `Dynamics.dynamics` (EGNN) module:
 *   embedding.weight, embedding.bias, embedding_out.weight, embedding_out.bias, then per block i:
 *   gcl_0.edge_mlp.0.{weight,bias}, gcl_0.edge_mlp.2.{weight,bias}, gcl_0.node_mlp.0.{weight,bias},
 *   gcl_0.node_mlp.2.{weight,bias}, (same for gcl_1 .. gcl_{inv_sublayers-1}),
 *   gcl_equiv.coord_mlp.0.{weight,bias}, gcl_equiv.coord_mlp.2.{weight,bias}, gcl_equiv.coord_mlp.4.weight */
int32_t dl_model_create(const dl_config* cfg, const float* const* weights, int32_t n_tensors, dl_model** out);
void dl_model_destroy(dl_model* m);

/* Largest number of REAL atoms per molecule the LDS-resident fully-connected kernels take with ONE workgroup per molecule
 * (a team of workgroups: dl_team_max_atoms). */
int32_t dl_max_atoms(void);

/* Dynamics.forward, fully-connected graph (src/egnn.py:374-447).
 *   xh          device [B,N,3+nf]   noisy state z_t (masked by node_mask inside, like the reference)
 *   t           device [B] (t_is_scalar=0) or [1] (t_is_scalar=1, the numel==1 branch :397-399)
 *   node_mask   device int8 [B,N]
 *   linker_mask device f32 [B,N] or NULL (NULL = no masking of the coordinate update, :113-114); the coordinate head
 *               (EquivariantUpdate, :101-125) is evaluated only for receiving atoms with linker_mask != 0: the
 *               reference multiplies every other atom's sum by zero (:113-116), so the outputs are the same
 *   edge_mask   device int8 [B,N,N] ({0,-1,-2} from collate; multiplies every message as-is) or NULL
 *               contract: edge_mask must be 0 wherever an endpoint has node_mask 0 (datasets.py:366-369).  A zero byte
 *               BETWEEN TWO REAL ATOMS (collate never writes one: every real pair is -1, the diagonal -2) removes that
 *               message like the reference's `* edge_mask` does, exactly in DL_PRECISION_FP32 and in the coordinate head of
 *               every mode; the f16 modes' GCL messages fold the mask into the SiLU reciprocal and apply such a byte as a
 *               factor 2^-100 instead of 0 - below one ulp of any fp32 sum while the message's pre-activation stays under
 *               the f16-range limit of bit4 below (2^75): not observable, but not bit-zero either
 *   context     device f32 [B,N,ctx] or NULL when ctx == 0
 *   out         device [B,N,3+nf]   eps_hat = cat[vel, h_final]; padded rows are written as 0
 *   nan_flags   device int32 [B]    bit0: NaN in vel, bit1: NaN in h_final, bit2: too many atoms
 *                                   (the caller raises FoundNaNException, src/egnn.py:441-442);
 *                                   bit4 (f16 modes only, always with bit0 | bit1): a magnitude bound of this molecule's
 *                                   activations reached 2^75 (3.8e22) - beyond the scales' range the fp16 operands would
 *                                   saturate silently, so `out` is void; DL_PRECISION_FP32 has no such limit.  The radius-graph
 *                                   kernels scale per tile, not per molecule: they report every molecule of the call */
int32_t dl_egnn_forward_fc(const dl_model* m, int32_t B, int32_t N,
                           const float* xh, const float* t, int32_t t_is_scalar,
                           const int8_t* node_mask, const float* linker_mask, const int8_t* edge_mask,
                           const float* context, float* out, int32_t* nan_flags,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Scratch of dl_egnn_forward_fc / dl_egnn_forward_fc_team / dl_sample_chain_fc for a batch of B molecules with `team`
 * compute units per molecule (0 or 1: one): per workgroup the node features and the pre-computed half of the node MLP that
 * cross the O(n^2) edge passes through L2 (124 KB: the T0 and residual tiles in accumulator order, the h fragment rows of a
 * 56..110-atom molecule across a coordinate pass), plus, for team > 1, the exchange buffers (116 KB per molecule) and arrival words. */
size_t dl_workspace_bytes(int32_t B, int32_t team);

/* DynamicsWithPockets.forward (src/egnn.py:470-552): radius graph rebuilt on the GPU every call
 * (ligand-ligand fully connected, pocket-pocket <= 4 A, ligand-pocket <= 10 A [4 A for 'FC-4A'], everything
 * <= 4 A for '4A'; no self loops; :554-596), EGNN with edge_mask = None.
 *   graph_type  0: '4A', 1: 'FC-4A', 2: 'FC-10A-4A'
 *   linker_mask device f32 [B,N] (required: it defines the ligand atoms together with context[..., -2])
 *   context     device f32 [B,N,ctx], last two channels = fragment-only / pocket-only masks (:486-487)
 *   workspace   device scratch of at least dl_pocket_workspace_bytes(B, N) bytes, caller-owned
 * Molecule membership is positional (atom v belongs to molecule v / N), which is what the reference's batch-index
 * "edge_mask" vector encodes (src/datasets.py:359-364).  Both precisions are supported. */
size_t dl_pocket_workspace_bytes(int32_t B, int32_t N);
int32_t dl_egnn_forward_pocket(const dl_model* m, int32_t B, int32_t N, int32_t graph_type,
                               const float* xh, const float* t, int32_t t_is_scalar,
                               const int8_t* node_mask, const float* linker_mask, const float* context,
                               float* out, int32_t* nan_flags, void* workspace, size_t workspace_bytes,
                               void* stream);

/* Dynamics.forward (src/egnn.py:374-447) for fully-connected graphs of ANY size: the HBM-resident per-pass kernels of
 * the pocket path run on the reference's own dense edge list — every pair whose int8 edge_mask value is non-zero, the
 * diagonal included (value -2, src/datasets.py:366-369), each message weighted by that value.  Use it for batches with a
 * molecule of more than dl_max_atoms() atoms (dl_egnn_forward_fc flags those with bit 2); several times slower than
 * the LDS-resident kernel.  Arguments as dl_egnn_forward_fc (edge_mask required, linker_mask may be NULL);
 * workspace: dl_pocket_workspace_bytes(B, N). */
int32_t dl_egnn_forward_fc_large(const dl_model* m, int32_t B, int32_t N, const float* xh, const float* t,
                                 int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask,
                                 const int8_t* edge_mask, const float* context, float* out, int32_t* nan_flags,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* Per-step scalars of the reverse process, computed by the host exactly as the reference does
 * (src/edm.py:180-185,199,202): one row per reverse step, in execution order (s = T-1 ... 0). */
typedef struct dl_step_coef {
    float t;            /* time feature (s+1)/T fed to the denoiser        */
    float alpha_ts;     /* alpha_{t|s}                                      */
    float c_eps;        /* sigma2_{t|s} / alpha_{t|s} / sigma_t             */
    float sigma;        /* sigma_{t|s} * sigma_s / sigma_t                  */
} dl_step_coef;

/* Fused tail of sample_p_zs_given_zt_only_linker (src/edm.py:196-206):
 *   z_s = z_t*frag + ((z_t/alpha_ts - c_eps*(eps_hat*lm)) + sigma*(noise*lm)) * lm
 * z_t, eps_hat, noise, z_s: device [B,N,D]; fragment_mask/linker_mask: device f32 [B,N]. */
int32_t dl_sampler_step(int32_t B, int32_t N, int32_t D, const float* z_t, const float* eps_hat,
                        const float* noise, const float* fragment_mask, const float* linker_mask,
                        dl_step_coef coef, float* z_s, void* stream);

/* EDM.sample_chain (src/edm.py:126-176) as ONE launch: every molecule runs its T reverse steps
 * and the final decode on one compute unit with its state resident in LDS. */
typedef struct dl_chain_args {
    int32_t B, N, T, keep_frames;
    const float* x;             /* device [B,N,3]   fragment-centred coordinates               */
    const float* h;             /* device [B,N,nf]  one-hot atom types (un-normalised)         */
    const int8_t* node_mask;    /* device [B,N]                                                */
    const float* fragment_mask; /* device [B,N]                                                */
    const float* linker_mask;   /* device [B,N]                                                */
    const int8_t* edge_mask;    /* device [B,N,N]                                              */
    const float* context;       /* device [B,N,ctx]                                            */
    const float* noise_x;       /* device [T+2,B,N,3]  standard normal draws, reference order: */
    const float* noise_h;       /* device [T+2,B,N,nf] draw 0 = initial z, 1..T = steps, T+1 = decode */
                                /* both NULL: the draws are generated inside the kernel (dl_philox_fill's stream) */
    uint64_t noise_seed;        /* key of the in-kernel generator                                          */
    int32_t mol_offset;         /* global index of molecule 0 of this batch (shards of one logical batch)  */
    int32_t team;               /* compute units per molecule: 0 or 1 = one (default); 2, 4 or 8 = a team, see below */
    const dl_step_coef* coefs;  /* device [T]       execution order (s = T-1 first)            */
    float inv_alpha0, sigma0, sigma_x;      /* final decode scalars (src/edm.py:213-216,237-242) */
    float norm_x, norm_h, bias_h;           /* norm_values[0], norm_values[1], norm_biases[1]    */
    float* chain;               /* device [keep_frames,B,N,3+nf]; frame 0 = final [x, one_hot(h)] */
    int32_t* nan_flags;         /* device [B]  bit0/bit1 as above (first offending forward only) */
    int32_t* nan_step;          /* device [B]  forward index (0..T) at which the flag was raised, or -1 */
    const int32_t* order;       /* device [B] or NULL: workgroup k samples molecule order[k].  One molecule occupies one
                                 * compute unit for the whole chain and workgroups are dispatched in index order, so a
                                 * batch larger than the chip finishes sooner when the big molecules go first
                                 * (longest-processing-time order); results are written at the molecule's own index. */
    void* workspace;            /* device scratch of dl_workspace_bytes(B, team) bytes, 16-byte aligned */
    size_t workspace_bytes;
    const int32_t* mol_index;   /* device [B] or NULL: entry b of this batch is molecule mol_offset + mol_index[b] of the
                                 * logical batch (NULL: mol_offset + b) - the key of the in-kernel noise; lets a caller
                                 * sample a non-contiguous part of a batch (e.g. only the molecules that fit this kernel) */
    int32_t order_first, order_count;   /* (ABI v7) this launch samples the molecules order[order_first .. order_first + order_count)
                                 * only (order_count = 0: all B; otherwise `order` must be given): every array keeps the extent
                                 * and the indexing of the whole batch, the workspace is sized for order_count molecules.  Lets a
                                 * caller put the few molecules beyond one-per-compute-unit on TEAMS in a second launch on another
                                 * stream - they take the compute units the smallest molecules of the first launch leave early -
                                 * instead of waiting for a whole second round (EDM.sample_chain, batches of 257..320 on 256 CUs) */
    /* (ABI v7) a chain in TWO launches - the static hand-over of compute units inside a ragged batch: molecule b runs the denoiser
     * calls q_begin[b] .. q_end[b]-1 of the T+1 (NULL: 0 / T+1).  A launch that stops a molecule early (q_end[b] <= T) writes its
     * state z (normalised, fp32, exactly as the kernel holds it) to z_state[b]; a launch that resumes one (q_begin[b] > 0) starts
     * from there instead of from x, h and draw 0.  The noise is a function of (molecule, atom, draw) - resuming needs no generator
     * state; frames are written by step index, frame 0 by the launch that runs the decode.  skip_flags (NULL or device [B]): a
     * molecule with a non-zero word is left alone (it ended - NaN - in the first launch).  EDM.sample_chain: the small molecules
     * of a batch finish in the first launch, the big ones stop where the small ones end and finish on TEAMS OF TWO in the second,
     * which uses the compute units the small ones left. */
    const int32_t* q_begin;
    const int32_t* q_end;
    float* z_state;             /* device [B,N,3+nf]; required when q_begin or q_end is given */
    const int32_t* skip_flags;
} dl_chain_args;

int32_t dl_sample_chain_fc(const dl_model* m, const dl_chain_args* args, void* stream);

/* Teams.  A batch smaller than the chip leaves compute units idle when every molecule sits on one of them (the reference's
 * default sampling batch is 64, generate.py:145), and a molecule of more than dl_max_atoms() atoms does not fit one compute
 * unit's LDS at all.  With team = 2, 4 or 8 that many workgroups share a molecule: its atoms are dealt round-robin, a member
 * keeps the state of its own atoms only and runs the per-atom phases (node MLP, projections, sampler algebra) for them alone;
 * its O(n^2) pair loops take its own atoms as receivers and every atom as sender, for which the members exchange the sender
 * rows and coordinates of their atoms once per pass through the workspace (release / acquire hand-off inside the launch,
 * placement-independent).  A team takes molecules of up to dl_team_max_atoms(team) = 110 atoms.
 * All team * ceil(B / 8) * 8 workgroups must be resident at once: dl_team_max(B) is the largest team the current device
 * holds for a batch of B (1, 2, 4 or 8), a larger request returns DL_ERR_BAD_ARG, and the launch is cooperative (the runtime
 * rejects a grid the device cannot hold).  Results agree with team = 1 to fp32 rounding (the order in which an atom's
 * messages are summed depends on the team size) and are bitwise repeatable for a given team size.
 * nan_flags bit 3: the members of a team did not all show up within the spin limit (another kernel held compute units for
 * seconds); every member gives up together, the sample is void and the caller re-runs the batch with team = 1 (or
 * dl_egnn_forward_fc_large).  With team > 1 the entry points zero nan_flags (and set nan_step to -1) on `stream` themselves. */
int32_t dl_team_max(int32_t B);
int32_t dl_team_max_atoms(int32_t team);
#ifdef DL_TEST_HOOKS
/* TEST BUILDS ONLY (-DDL_TEST_HOOKS: difflinker_amd/libdifflinker_hip_testhooks.so, built beside the product library and loaded
 * by the fault-injection tests alone; the product library does not export it).  Member 1 of every team of the NEXT `launches`
 * team launches gives up at its first exchange (exercises the fail-together path above); the count runs down by itself - the
 * switch cannot stay on by accident - and 0 clears it */
void dl_debug_team_fault(int32_t launches);
#endif
/* dl_egnn_forward_fc with a team per molecule (team = 1: identical to dl_egnn_forward_fc) */
int32_t dl_egnn_forward_fc_team(const dl_model* m, int32_t B, int32_t N, const float* xh, const float* t,
                                int32_t t_is_scalar, const int8_t* node_mask, const float* linker_mask,
                                const int8_t* edge_mask, const float* context, float* out, int32_t* nan_flags,
                                int32_t team, void* workspace, size_t workspace_bytes, void* stream);

/* One step of InpaintingEDM.sample_chain after the denoiser call (src/edm.py:568-596), or its final decode
 * (:599-610, :674-713), for one batch: the linker atoms take the p(z_s|z_t) sample, the fragment atoms are re-drawn from
 * q(z_s|z_t,x), the position noise is centre-of-gravity free (utils.py:158-168), the denoiser's velocity is centred
 * (egnn.py:444-445) and the centre of gravity of z_s is projected out. */
typedef struct dl_inpaint_coef {
    float alpha_ts, c_eps, sigma;           /* p: mu = z/alpha_ts - c_eps*eps_hat;  sigma of both draws   */
    float a_q, b_q;                         /* q: mu = a_q*z + b_q*(xh*fragment_mask)                     */
    int32_t decode;                         /* 0: reverse step, 1: final decode                           */
    float inv_alpha0, sigma0, sigma_x;      /* decode scalars                                             */
    float norm_x, norm_h, bias_h;           /* decode: un-normalisation, then one-hot of the features     */
} dl_inpaint_coef;
/*   z_t, eps_hat, xh_frag, z_s  device f32 [B,N,3+nf]   (eps_hat: raw denoiser output, centred here)
 *   noise_p*, noise_q*          device f32 [B,N,3] / [B,N,nf]: the four torch.randn draws of the step, unmasked
 *   node_mask, fragment_mask, linker_mask  device f32 [B,N] */
int32_t dl_inpaint_step(int32_t B, int32_t N, int32_t nf, const float* z_t, const float* eps_hat, const float* xh_frag,
                        const float* noise_px, const float* noise_ph, const float* noise_qx, const float* noise_qh,
                        const float* node_mask, const float* fragment_mask, const float* linker_mask,
                        dl_inpaint_coef coef, float* z_s, void* stream);

/* The in-kernel noise stream as a bank (for host-driven loops and tests): Philox4x32-10, key = seed, counter =
 * (mol_offset + m(b), atom position n, draw0 + k, component / 4) with m(b) = mol_index[b] (device int32 [B]) or b when mol_index
 * is NULL - the same keying as dl_chain_args.mol_offset / mol_index, so a part of a batch gets exactly its own rows;
 * four outputs -> four standard normals by Box-Muller; component d < 3 is noise_x[k][b][n][d], d >= 3 is noise_h[k][b][n][d-3].
 * Independent of the batch split. */
int32_t dl_philox_fill(uint64_t seed, int32_t mol_offset, const int32_t* mol_index, int32_t B, int32_t N, int32_t nf, int32_t draw0,
                       int32_t n_draws, float* noise_x, float* noise_h, void* stream);

/* Diagnostics (libraries built with -DDL_PROFILE only; dl_profile_max_events() returns 0 otherwise): when set
 * (device uint64 [8 waves][dl_profile_max_events()][2], or NULL to disable), the first workgroup of the next
 * launches logs (phase tag, shader clock) pairs of its first forward. */
void dl_set_profile_buffer(void* device_buf);
int32_t dl_profile_max_events(void);

/* ---- linker-size predictor -------------------------------------------------------------------------
 * SizeGNN hyper-parameters (src/linker_size.py:46); hidden_nf must be 128.  nn.BatchNorm1d
 * (normalization='batch_norm') is an affine map in eval mode: the caller folds it into the adjacent
 * Linear before handing the tensors over. */
typedef struct dl_size_config {
    int32_t in_node_nf;
    int32_t hidden_nf;
    int32_t out_node_nf;
    int32_t n_layers;
} dl_size_config;

typedef struct dl_size_model dl_size_model;

/* tensors (host fp32, nn.Linear [out,in] row-major), 4 + 8 * n_layers of them:
 *   embedding_in.weight, .bias,
 *   per GCL (gcl1, gcl_layers.0, ...): edge_mlp.0.weight [128,257], .bias, edge_mlp.2.weight, .bias,
 *                                      node_mlp.0.weight [128,256], .bias, node_mlp.<last>.weight, .bias,
 *   embedding_out.weight [out,128], .bias */
int32_t dl_size_model_num_tensors(const dl_size_config* cfg);
int32_t dl_size_model_create(const dl_size_config* cfg, const void* const* tensors, int32_t n_tensors,
                             dl_size_model** out);
void dl_size_model_destroy(dl_size_model* m);
int32_t dl_size_max_fragment_atoms(void);

/* logits[b] = mean over the N padded nodes of embedding_out(GCL^n(embedding_in(one_hot * fragment_mask)))
 *   one_hot        device f32 [B,N,in_node_nf]
 *   positions      device f32 [B,N,3]   (may be NULL when `distances` is given)
 *   fragment_mask  device f32 [B,N]     node mask of the GNN (fragment atoms; 'fragment_only_mask' with pockets)
 *   edge_mask      device f32 [B,N,N]   e = b*N*N + i*N + j; an edge is kept where edge_mask != 0 and, when
 *                                       `distances` is NULL, the squared distance |x_i - x_j|^2 < 6
 *                                       (src/linker_size_lightning.py:106-107: coord2diff returns the SQUARED
 *                                       distance and that is what is compared with 6 and fed to the edge MLP)
 *   distances      device f32 [B,N,N] or NULL: precomputed edge attribute (SizeGNN.forward's own signature,
 *                                       src/linker_size.py:83); edge_mask is then taken as final
 *   logits         device f32 [B,out_node_nf]
 *   flags          device int32 [B]     bit0: NaN in the logits, bit2: more fragment atoms than
 *                                       dl_size_max_fragment_atoms() */
int32_t dl_size_gnn_forward(const dl_size_model* m, int32_t B, int32_t N, const float* one_hot,
                            const float* positions, const float* fragment_mask, const float* edge_mask,
                            const float* distances, float* logits, int32_t* flags, void* stream);


const char* dl_error_string(int32_t status);
int32_t dl_last_hip_error(void);
int32_t dl_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DIFFLINKER_HIP_H */
