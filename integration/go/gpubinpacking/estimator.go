package gpubinpacking

/*
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"unsafe"

	apiv1 "k8s.io/api/core/v1"
	"k8s.io/autoscaler/cluster-autoscaler/cloudprovider"
	"k8s.io/autoscaler/cluster-autoscaler/estimator"
	"k8s.io/autoscaler/cluster-autoscaler/simulator/clustersnapshot"
	"k8s.io/autoscaler/cluster-autoscaler/simulator/framework"
)

// DeviceLimiter is what the shim needs from the limiter beyond estimator.EstimationLimiter: the node limit StartEstimation
// computed (thresholdBasedEstimationLimiter.maxNodes is private, threshold_based_limiter.go:27-43; builder.go's deviceLimiter folds
// the public thresholds with the reference's rule).  The duration limit stays on the host (checked between calls; the device
// runs with the node limit only).
type DeviceLimiter interface {
	estimator.EstimationLimiter
	MaxNodes() int // <0 forbid, 0 unlimited, >0 cap: casim_groups.max_nodes verbatim
}

type gpuEstimator struct {
	engine   *Engine
	snapshot clustersnapshot.ClusterSnapshot
	limiter  DeviceLimiter
	context  estimator.EstimationContext
	fastpath bool
	shared   *Shared             // nil = per-call mode only
	fallback estimator.Estimator // the reference's BinpackingNodeEstimator: groups outside the encoded predicate subset
}

// New is what NewEstimatorBuilder (builder.go) returns for every (snapshot, context) pair.
func New(engine *Engine, snapshot clustersnapshot.ClusterSnapshot, limiter DeviceLimiter, context estimator.EstimationContext,
	fastpath bool, shared *Shared, fallback estimator.Estimator) estimator.Estimator {
	if engine == nil {
		return fallback
	}
	return &gpuEstimator{engine: engine, snapshot: snapshot, limiter: limiter, context: context, fastpath: fastpath, shared: shared, fallback: fallback}
}

// Estimate implements estimator.Estimator (estimator.go:53-56) with the semantics of BinpackingNodeEstimator.Estimate
// (binpacking_estimator.go:102-161): node count, and the pods that fit in placement order.
func (g *gpuEstimator) Estimate(pegs []estimator.PodEquivalenceGroup, tmpl *framework.NodeInfo, ng cloudprovider.NodeGroup) (int, []*apiv1.Pod) {
	observeBinpackingHeterogeneity(pegs, tmpl) // the metric stays on the Go side (binpacking_estimator.go:107)
	g.limiter.StartEstimation(pegs, ng, g.context)
	defer g.limiter.EndEstimation()
	maxNodes := g.limiter.MaxNodes()
	existing := nodeCount(g.snapshot)

	// ---- prefetch mode: the batch of this loop may hold the answer (prefetch.go) ----
	if g.shared != nil {
		if r, order, placed, ok := g.shared.lookup(ng, tmpl, pegs, maxNodes, existing); ok {
			if r.status == C.CASIM_NG_OK {
				return int(r.node_count), prefixPods(pegs, order, placed)
			}
			return g.fallback.Estimate(pegs, tmpl, ng) // the batch delegated this group (CASIM_NG_UNSUPPORTED)
		}
	}

	// ---- per-call mode: ONE casim_estimate_batch with one group record and the PEGs the orchestrator passed ----
	s := newSession()
	defer s.close()
	ids := make([]C.int32_t, len(pegs))
	for i, p := range pegs {
		ids[i] = s.peg(p)
	}
	s.group(tmpl, maxNodes, existing, g.lastIndex(), ids)
	pt, gt, err := s.tables()
	if err != nil {
		return g.fallback.Estimate(pegs, tmpl, ng)
	}
	n := len(pegs)
	var nodeCount, podsScheduled, nodesAdded, limiterNodes, lastIndexOut, status C.int32_t
	var cpu, mem C.int64_t
	order := make([]C.int32_t, n+1)
	placed := make([]C.int32_t, n+1)
	res := C.casim_results{node_count: &nodeCount, pods_scheduled: &podsScheduled, nodes_added: &nodesAdded, limiter_nodes: &limiterNodes,
		last_index_out: &lastIndexOut, status: &status, req_cpu_sum: &cpu, req_mem_sum: &mem, order: &order[0], placed: &placed[0]}
	var opts C.casim_options
	if g.fastpath {
		opts.fastpath = 1
	}
	g.engine.mu.Lock()
	rc := C.casim_estimate_batch(g.engine.ctx, &pt, &gt, &opts, &res)
	g.engine.mu.Unlock()
	if rc != C.CASIM_OK || status != C.CASIM_NG_OK {
		return g.fallback.Estimate(pegs, tmpl, ng) // fail closed: error, or a predicate outside the encoded subset
	}
	g.setLastIndex(int(lastIndexOut)) // the runner's lastIndex persists across Estimates (plugin_runner.go:33-36,138)
	return int(nodeCount), prefixPods(pegs, order[:n], placed[:n])
}

// prefixPods rebuilds Estimate()'s []*Pod: PEG order[k] was processed k-th and placed[k] of its pods were scheduled — always
// a prefix of the PEG (identical pods, SURVEY N10).
func prefixPods(pegs []estimator.PodEquivalenceGroup, order, placed []C.int32_t) []*apiv1.Pod {
	total := 0
	for _, p := range placed {
		total += int(p)
	}
	pods := make([]*apiv1.Pod, 0, total)
	for k, id := range order {
		pods = append(pods, pegs[id].Pods[:placed[k]]...)
	}
	return pods
}

func nodeCount(s clustersnapshot.ClusterSnapshot) int {
	infos, err := s.ListNodeInfos()
	if err != nil {
		return 0
	}
	return len(infos)
}

// lastIndex / setLastIndex: the shim never runs the snapshot's SchedulerPluginRunner for simulated pods, so it keeps the
// runner's lastIndex (scheduling_opts.go:39-63) itself, per snapshot, shared by the estimators of a loop.
func (g *gpuEstimator) lastIndex() int {
	if g.shared != nil {
		return g.shared.lastIndex
	}
	return 0
}
func (g *gpuEstimator) setLastIndex(v int) {
	if g.shared != nil {
		g.shared.lastIndex = v
	}
}

var _ = unsafe.Pointer(nil)
