package gpubinpacking

/*
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"hash/fnv"
	"runtime"
	"unsafe"

	apiv1 "k8s.io/api/core/v1"
	"k8s.io/autoscaler/cluster-autoscaler/cloudprovider"
	ca_context "k8s.io/autoscaler/cluster-autoscaler/context"
	coreoptions "k8s.io/autoscaler/cluster-autoscaler/core/options"
	"k8s.io/autoscaler/cluster-autoscaler/core/scaleup/equivalence"
	"k8s.io/autoscaler/cluster-autoscaler/estimator"
	"k8s.io/autoscaler/cluster-autoscaler/processors/nodegroups"
	"k8s.io/autoscaler/cluster-autoscaler/simulator/framework"
)

// Shared is what the NodeGroupListProcessor wrapper and the estimators of one ScaleUp loop have in common: the prefetch
// cache (casim_prefetch_*, csrc/casim_prefetch.cpp) and the runner's lastIndex as the shim models it.
type Shared struct {
	engine        *Engine
	cache         *C.casim_prefetch
	limiter       DeviceLimiter
	maxNodesTotal int
	fastpath      bool
	// the tables of the current loop (every PEG, every candidate group), kept until the next fill: a lookup that misses is served from
	// THEM — one casim_enc_group_rows + one casim_estimate_batch, no encoder work per Estimate() (estimator.go: estimateOnLoopTables)
	sess     *session
	pegs     C.casim_pegs
	pegID    map[*apiv1.Pod]C.int32_t // exemplar pod -> PEG id of the loop's tables
	groupRow map[C.uint64_t]C.int32_t // groupKey -> row of the loop's group table
	// the batch of the loop in its order (= the order of Estimate() calls the chain assumes): rows of the loop's group table, their keys, the
	// PEG keys — what a re-chain needs (rechain below); rechains counts them per loop
	batchRows []C.int32_t
	batchKeys []C.uint64_t
	pegKeys   []C.uint64_t
	rechains  int
	loopLastIndex int // the snapshot runner's lastIndex (estimator.go: runnerState.get) when the batch of the current loop was filled: the FIRST group of the batch starts from it
	runners       *Runners // the builder's registry (builder.go sets it)
	// Unchained = the round-4 protocol: every group of the batch starts from loopLastIndex and hits never move the runner.  Default (false): the
	// batch hands lastIndex from group to group (casim_options.chain_last_index), the way the orchestrator's loop does on one snapshot
	// (plugin_runner.go:138); over BASELINE configs C1-C4 and 400 fuzz scenarios the two differ in the (node count, pods) of 6 of 2381 groups
	// (profiles/r10_chain_rate.json) — rarely, but the chained batch is the reference's answer.
	Unchained bool
}

// EstimatorName is the --estimator value that selects this package (autoscaler_go.patch).
const EstimatorName = "gpu-binpacking"

// NewShared is created once, next to the Engine; NewEstimatorBuilder hands it the limiter.
func NewShared(engine *Engine, maxNodesTotal int, fastpath bool) *Shared {
	return &Shared{engine: engine, cache: C.casim_prefetch_create(engine.ctx), maxNodesTotal: maxNodesTotal, fastpath: fastpath}
}

// SimilarNodeGroups returns what ComputeSimilarNodeGroups would find for a group (orchestrator.go:395): the configured
// NodeGroupSetProcessor when BalanceSimilarNodeGroups is on, else nothing.
func SimilarNodeGroups(opts *coreoptions.AutoscalerOptions) func(*ca_context.AutoscalingContext, cloudprovider.NodeGroup, map[string]*framework.NodeInfo) []cloudprovider.NodeGroup {
	return func(ctx *ca_context.AutoscalingContext, ng cloudprovider.NodeGroup, infos map[string]*framework.NodeInfo) []cloudprovider.NodeGroup {
		if !opts.BalanceSimilarNodeGroups || opts.Processors.NodeGroupSetProcessor == nil {
			return nil
		}
		similar, err := opts.Processors.NodeGroupSetProcessor.FindSimilarNodeGroups(ctx, ng, infos)
		if err != nil {
			return nil
		}
		return similar
	}
}

// Close releases the cache and the loop's tables.
func (s *Shared) Close() {
	s.dropLoopTables()
	C.casim_prefetch_destroy(s.cache)
}

func (s *Shared) dropLoopTables() {
	if s.sess != nil {
		s.sess.close()
		s.sess = nil
	}
	s.pegID, s.groupRow = nil, nil
	s.batchRows, s.batchKeys, s.pegKeys, s.rechains = nil, nil, nil, 0
}

// Keys are opaque to libcasim.  A PEG is its exemplar pod (the orchestrator builds the groups once per loop and passes the same
// *apiv1.Pod values to SchedulablePodGroups and to every Estimate); a node group is its Id() plus the identity of the template
// NodeInfo the loop uses for it.
func pegKey(g estimator.PodEquivalenceGroup) C.uint64_t {
	return C.uint64_t(uintptr(unsafe.Pointer(g.Exemplar())))
}
func groupKey(ng cloudprovider.NodeGroup, tmpl *framework.NodeInfo) C.uint64_t {
	h := fnv.New64a()
	h.Write([]byte(ng.Id()))
	return C.uint64_t(h.Sum64() ^ uint64(uintptr(unsafe.Pointer(tmpl)))*0x9E3779B97F4A7C15)
}

// fill runs ONE casim_estimate_batch over every PEG and every candidate node group; SchedulablePodGroups is derived on the
// device (peg_offsets == NULL), max_nodes per group comes from the reference's own limiter.
func (s *Shared) fill(autoscalingCtx *ca_context.AutoscalingContext, pegs []estimator.PodEquivalenceGroup, ngs []cloudprovider.NodeGroup,
	infos map[string]*framework.NodeInfo, similar func(cloudprovider.NodeGroup) []cloudprovider.NodeGroup) error {
	C.casim_prefetch_clear(s.cache)
	s.dropLoopTables()
	if len(pegs) == 0 || len(ngs) == 0 || s.limiter == nil {
		return nil
	}
	sess := newSession()
	keep := false
	defer func() {
		if !keep {
			sess.close()
		}
	}()
	pkeys := make([]C.uint64_t, len(pegs))
	pegID := make(map[*apiv1.Pod]C.int32_t, len(pegs))
	ids, err := sess.pegs(pegs) // (one crossing for the exemplars: casim_enc_add_pods)
	if err != nil {
		return err
	}
	for i, p := range pegs {
		pegID[p.Exemplar()] = ids[i]
		pkeys[i] = pegKey(p)
	}
	existing := nodeCount(autoscalingCtx.ClusterSnapshot)
	if s.runners == nil {
		s.runners = NewRunners()
	}
	s.loopLastIndex = s.runners.of(autoscalingCtx.ClusterSnapshot).get() // (the snapshot's own runner when it implements RunnerIndex)
	gkeys := make([]C.uint64_t, 0, len(ngs))
	rows := make([]C.int32_t, 0, len(ngs))
	groupRow := make(map[C.uint64_t]C.int32_t, len(ngs))
	for _, ng := range ngs {
		tmpl, ok := infos[ng.Id()]
		if !ok {
			continue
		}
		// estimatorBuilder(snapshot, NewEstimationContext(MaxNodesTotal, SimilarNodeGroups, currentNodeCount)) — orchestrator.go:409-412
		ectx := estimator.NewEstimationContext(s.maxNodesTotal, similar(ng), existing)
		s.limiter.StartEstimation(pegs, ng, ectx)
		row := sess.group(tmpl, s.limiter.MaxNodes(), existing, s.loopLastIndex, nil)
		s.limiter.EndEstimation()
		gkeys = append(gkeys, groupKey(ng, tmpl))
		rows = append(rows, row)
		groupRow[groupKey(ng, tmpl)] = row
	}
	if len(gkeys) == 0 { // no candidate group came with a template: nothing to prefetch (and no &gkeys[0] to take)
		return nil
	}
	pt, gt, err := sess.tables()
	if err != nil {
		return err
	}
	var opts C.casim_options
	if s.fastpath {
		opts.fastpath = 1
	}
	if !s.Unchained {
		opts.chain_last_index = 1 // groups in the order the orchestrator will call Estimate(): the order of `ngs`
	}
	s.engine.mu.Lock()
	defer s.engine.mu.Unlock()
	if err := rcErr(C.casim_prefetch_fill(s.cache, &pt, &gt, &opts, &gkeys[0], &pkeys[0]), "casim_prefetch_fill"); err != nil {
		return err
	}
	keep = true // the encoder stays: its tables serve the per-call path of this loop
	s.sess, s.pegs, s.pegID, s.groupRow = sess, pt, pegID, groupRow
	s.batchRows, s.batchKeys, s.pegKeys, s.rechains = rows, gkeys, pkeys, 0
	return nil
}

// maxRechains bounds the batches one loop may spend on re-chaining (each is one device call over the REST of the loop's groups).
const maxRechains = 4

// rechain: a chained batch answers group i for the lastIndex group i - 1 left.  When the chain is LEFT — a group ran on the reference path
// (CASIM_NG_UNSUPPORTED, a routed small call) or was skipped, so the runner stands elsewhere — every later lookup of the loop misses on
// lastIndex alone (CASIM_PREFETCH_MISS_LAST_INDEX) and would take one per-call device trip each (ADVICE r5).  Instead the REST of the
// batch, from the group that missed on, is estimated again as ONE chained batch that starts from the runner's lastIndex of now: the rows
// of the loop's group table (casim_enc_group_rows, no encoding), their limits as the fill computed them, the first group's last_index
// overridden.  The cache then holds the rest of the loop; groups answered before are gone (a second Estimate for one of them misses on
// the group and takes the per-call path on the loop's tables).  false: nothing to re-chain (unknown group, last group of the batch, the
// budget of this loop is spent, an error) — the caller continues with the per-call path.
func (s *Shared) rechain(ng cloudprovider.NodeGroup, tmpl *framework.NodeInfo, lastIndex int) bool {
	if s.sess == nil || s.Unchained || s.rechains >= maxRechains {
		return false
	}
	key := groupKey(ng, tmpl)
	from := -1
	for i, k := range s.batchKeys {
		if k == key {
			from = i
			break
		}
	}
	if from < 0 || len(s.batchKeys)-from < 2 {
		return false
	}
	s.rechains++
	n := len(s.batchKeys) - from
	var rows C.casim_groups
	s.engine.mu.Lock()
	defer s.engine.mu.Unlock()
	if C.casim_enc_group_rows(s.sess.enc, &s.batchRows[from], C.int32_t(n), &rows) != C.CASIM_OK {
		return false
	}
	li := make([]C.int32_t, n) // (the chain reads the FIRST group's entry; the others are ignored, casim.h casim_options.chain_last_index)
	for i := range li {
		li[i] = C.int32_t(lastIndex)
	}
	var pin runtime.Pinner
	defer pin.Unpin()
	pin.Pin(&li[0])
	rows.last_index = &li[0]
	var opts C.casim_options
	if s.fastpath {
		opts.fastpath = 1
	}
	opts.chain_last_index = 1
	return C.casim_prefetch_fill(s.cache, &s.pegs, &rows, &opts, &s.batchKeys[from], &s.pegKeys[0]) == C.CASIM_OK
}

// estimateOnLoopTables is the per-call path WITHOUT encoding: the group's row of the loop's tables (casim_enc_group_rows, n = 1), the
// PEG list the orchestrator passed as ids of the loop's PEG table, this call's limiter answer / E / lastIndex, one casim_estimate_batch.
// ok == false: the loop's tables do not know this group or one of the PEGs (the caller encodes the call by itself).
func (s *Shared) estimateOnLoopTables(ng cloudprovider.NodeGroup, tmpl *framework.NodeInfo, pegs []estimator.PodEquivalenceGroup,
	maxNodes, existing, lastIndex int, fastpath bool) (nodeCount int, order, placed []C.int32_t, lastIndexOut int, status C.int32_t, ok bool) {
	if s.sess == nil {
		return
	}
	row, found := s.groupRow[groupKey(ng, tmpl)]
	if !found {
		return
	}
	n := len(pegs)
	ids := make([]C.int32_t, n+1)
	for i := range pegs {
		id, known := s.pegID[pegs[i].Exemplar()]
		if !known || int(s.pegCount(id)) != len(pegs[i].Pods) {
			return
		}
		ids[i] = id
	}
	var rows C.casim_groups
	s.engine.mu.Lock()
	defer s.engine.mu.Unlock()
	if C.casim_enc_group_rows(s.sess.enc, &row, 1, &rows) != C.CASIM_OK {
		return
	}
	// Go memory referenced from C structs is pinned for the call (cgo pointer rules)
	var pin runtime.Pinner
	defer pin.Unpin()
	off := []C.int32_t{0, C.int32_t(n)}
	lim := []C.int32_t{C.int32_t(maxNodes), C.int32_t(existing), C.int32_t(lastIndex)}
	scal := make([]C.int32_t, 6)
	sums := make([]C.int64_t, 2)
	order = make([]C.int32_t, n+1)
	placed = make([]C.int32_t, n+1)
	for _, p := range []*C.int32_t{&off[0], &ids[0], &lim[0], &scal[0], &order[0], &placed[0]} {
		pin.Pin(p)
	}
	pin.Pin(&sums[0])
	rows.peg_offsets, rows.peg_index = &off[0], &ids[0]
	rows.max_nodes, rows.existing_nodes, rows.last_index = &lim[0], &lim[1], &lim[2]
	res := C.casim_results{node_count: &scal[0], pods_scheduled: &scal[1], nodes_added: &scal[2], limiter_nodes: &scal[3], last_index_out: &scal[4],
		status: &scal[5], req_cpu_sum: &sums[0], req_mem_sum: &sums[1], order: &order[0], placed: &placed[0]}
	var opts C.casim_options
	if fastpath {
		opts.fastpath = 1
	}
	if C.casim_estimate_batch(s.engine.ctx, &s.pegs, &rows, &opts, &res) != C.CASIM_OK {
		return
	}
	// order holds ids of the loop's PEG table: back to positions in the caller's list
	pos := make(map[C.int32_t]C.int32_t, n)
	for i := 0; i < n; i++ {
		pos[ids[i]] = C.int32_t(i)
	}
	for k := 0; k < n; k++ {
		order[k] = pos[order[k]]
	}
	return int(scal[0]), order[:n], placed[:n], int(scal[4]), scal[5], true
}

// pegCount reads the pod count of PEG id from the loop's tables (a hit needs the same pods, not only the same exemplar).
func (s *Shared) pegCount(id C.int32_t) C.int32_t {
	return *(*C.int32_t)(unsafe.Add(unsafe.Pointer(s.pegs.count), uintptr(id)*unsafe.Sizeof(C.int32_t(0))))
}

// lookup: a hit only when the call asks exactly the question the batch answered (group, PEG set, limiter answer, E, lastIndex).
// runnerLastIndex = the snapshot runner's lastIndex at the time of THIS Estimate(): the chained batch answered group i for the lastIndex
// group i - 1 left behind, so the call hits iff the runner stands there (the calls so far arrived in the batch's order).
func (s *Shared) lookup(ng cloudprovider.NodeGroup, tmpl *framework.NodeInfo, pegs []estimator.PodEquivalenceGroup, maxNodes, existing, runnerLastIndex int) (
	r C.casim_prefetch_result, order, placed []C.int32_t, ok bool) {
	li := runnerLastIndex
	if s.Unchained {
		li = s.loopLastIndex
	}
	n := len(pegs)
	keys := make([]C.uint64_t, n+1)
	for i, p := range pegs {
		keys[i] = pegKey(p)
	}
	order = make([]C.int32_t, n+1)
	placed = make([]C.int32_t, n+1)
	rc := C.casim_prefetch_lookup(s.cache, groupKey(ng, tmpl), &keys[0], C.int32_t(n), C.int32_t(maxNodes), C.int32_t(existing),
		C.int32_t(li), &r, &order[0], &placed[0])
	return r, order[:n], placed[:n], rc == C.CASIM_OK
}

// ---- the NodeGroupListProcessor wrapper --------------------------------------------------------------------------------
// ScaleUp hands processors.NodeGroupListProcessor.Process every candidate node group, every template NodeInfo and the pending
// pods BEFORE its two loops run (core/scaleup/orchestrator/orchestrator.go:121-123); AutoscalingProcessors.NodeGroupListProcessor is
// an injection point of AutoscalerOptions.Processors, like the EstimatorBuilder.
type prefetchProcessor struct {
	inner   nodegroups.NodeGroupListProcessor
	shared  *Shared
	similar func(*ca_context.AutoscalingContext, cloudprovider.NodeGroup, map[string]*framework.NodeInfo) []cloudprovider.NodeGroup
}

// WrapNodeGroupListProcessor: opts.Processors.NodeGroupListProcessor = gpubinpacking.WrapNodeGroupListProcessor(inner, shared, similar).
// `similar` is processors.NodeGroupSetProcessor.FindSimilarNodeGroups when BalanceSimilarNodeGroups is on (what
// ComputeSimilarNodeGroups calls, orchestrator.go:395), else a function returning nil.
func WrapNodeGroupListProcessor(inner nodegroups.NodeGroupListProcessor, shared *Shared,
	similar func(*ca_context.AutoscalingContext, cloudprovider.NodeGroup, map[string]*framework.NodeInfo) []cloudprovider.NodeGroup) nodegroups.NodeGroupListProcessor {
	return &prefetchProcessor{inner: inner, shared: shared, similar: similar}
}

func (p *prefetchProcessor) Process(autoscalingCtx *ca_context.AutoscalingContext, ngs []cloudprovider.NodeGroup, infos map[string]*framework.NodeInfo,
	pods []*apiv1.Pod) ([]cloudprovider.NodeGroup, map[string]*framework.NodeInfo, error) {
	ngs, infos, err := p.inner.Process(autoscalingCtx, ngs, infos, pods)
	if err != nil {
		return ngs, infos, err
	}
	// the orchestrator's own groups: BuildPodGroups is deterministic in WHICH pods share a group and in each group's first pod (the
	// exemplar = our key); only the order of the groups is a map's (groups.go:62-104), and the cache compares PEG lists as sets
	groups := equivalence.BuildPodGroups(pods)
	pegs := make([]estimator.PodEquivalenceGroup, 0, len(groups))
	for _, g := range groups {
		pegs = append(pegs, estimator.PodEquivalenceGroup{Pods: g.Pods})
	}
	if ferr := p.shared.fill(autoscalingCtx, pegs, ngs, infos, func(ng cloudprovider.NodeGroup) []cloudprovider.NodeGroup {
		return p.similar(autoscalingCtx, ng, infos)
	}); ferr != nil {
		C.casim_prefetch_clear(p.shared.cache) // the cache is an accelerator: every Estimate() of this loop takes the per-call path
	}
	return ngs, infos, nil
}

func (p *prefetchProcessor) CleanUp() { p.inner.CleanUp() }
