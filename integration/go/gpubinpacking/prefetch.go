package gpubinpacking

/*
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"hash/fnv"
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
	loopLastIndex int // the snapshot runner's lastIndex (estimator.go: runnerState) when the batch of the current loop was filled: every group of the batch starts from it
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

// Close releases the cache.
func (s *Shared) Close() { C.casim_prefetch_destroy(s.cache) }

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
	if len(pegs) == 0 || len(ngs) == 0 || s.limiter == nil {
		return nil
	}
	sess := newSession()
	defer sess.close()
	pkeys := make([]C.uint64_t, len(pegs))
	for i, p := range pegs {
		sess.peg(p)
		pkeys[i] = pegKey(p)
	}
	existing := nodeCount(autoscalingCtx.ClusterSnapshot)
	rs := runnerOf(autoscalingCtx.ClusterSnapshot)
	rs.mu.Lock()
	s.loopLastIndex = rs.lastIndex
	rs.mu.Unlock()
	gkeys := make([]C.uint64_t, 0, len(ngs))
	for _, ng := range ngs {
		tmpl, ok := infos[ng.Id()]
		if !ok {
			continue
		}
		// estimatorBuilder(snapshot, NewEstimationContext(MaxNodesTotal, SimilarNodeGroups, currentNodeCount)) — orchestrator.go:409-412
		ectx := estimator.NewEstimationContext(s.maxNodesTotal, similar(ng), existing)
		s.limiter.StartEstimation(pegs, ng, ectx)
		sess.group(tmpl, s.limiter.MaxNodes(), existing, s.loopLastIndex, nil)
		s.limiter.EndEstimation()
		gkeys = append(gkeys, groupKey(ng, tmpl))
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
	s.engine.mu.Lock()
	defer s.engine.mu.Unlock()
	return rcErr(C.casim_prefetch_fill(s.cache, &pt, &gt, &opts, &gkeys[0], &pkeys[0]), "casim_prefetch_fill")
}

// lookup: a hit only when the call asks exactly the question the batch answered (group, PEG set, limiter answer, E, lastIndex).
func (s *Shared) lookup(ng cloudprovider.NodeGroup, tmpl *framework.NodeInfo, pegs []estimator.PodEquivalenceGroup, maxNodes, existing int) (
	r C.casim_prefetch_result, order, placed []C.int32_t, ok bool) {
	n := len(pegs)
	keys := make([]C.uint64_t, n+1)
	for i, p := range pegs {
		keys[i] = pegKey(p)
	}
	order = make([]C.int32_t, n+1)
	placed = make([]C.int32_t, n+1)
	rc := C.casim_prefetch_lookup(s.cache, groupKey(ng, tmpl), &keys[0], C.int32_t(n), C.int32_t(maxNodes), C.int32_t(existing),
		C.int32_t(s.loopLastIndex), &r, &order[0], &placed[0])
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
