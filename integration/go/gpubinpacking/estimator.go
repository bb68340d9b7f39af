package gpubinpacking

/*
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"runtime"
	"strconv"
	"sync"

	apiv1 "k8s.io/api/core/v1"
	"k8s.io/autoscaler/cluster-autoscaler/cloudprovider"
	"k8s.io/autoscaler/cluster-autoscaler/estimator"
	"k8s.io/autoscaler/cluster-autoscaler/metrics"
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
	runner   *runnerState        // lastIndex of THIS snapshot's plugin runner, as the shim threads it (never nil)
	fallback estimator.Estimator // the reference's BinpackingNodeEstimator: groups outside the encoded predicate subset
}

// runnerState is the one piece of SchedulerPluginRunner state an Estimate reads and writes: lastIndexOrderMapping.lastIndex
// (cluster-autoscaler/simulator/clustersnapshot/scheduling_opts.go:39-63).  In the reference the runner lives inside the snapshot
// (predicate/predicate_snapshot.go:64) and its lastIndex survives every Estimate of every loop (plugin_runner.go:33-36,138); the
// orchestrator builds a fresh estimator per node group (orchestrator.go:409-413), so the shim keeps one runnerState per snapshot
// OUTSIDE the estimators — in prefetch mode and in per-call mode alike (the analyser path has no Shared).
type runnerState struct {
	mu        sync.Mutex
	lastIndex int
}

var runnerStates sync.Map // clustersnapshot.ClusterSnapshot -> *runnerState

func runnerOf(snapshot clustersnapshot.ClusterSnapshot) *runnerState {
	if st, ok := runnerStates.Load(snapshot); ok {
		return st.(*runnerState)
	}
	st, _ := runnerStates.LoadOrStore(snapshot, &runnerState{})
	return st.(*runnerState)
}

// New is what NewEstimatorBuilder (builder.go) returns for every (snapshot, context) pair.
func New(engine *Engine, snapshot clustersnapshot.ClusterSnapshot, limiter DeviceLimiter, context estimator.EstimationContext,
	fastpath bool, shared *Shared, fallback estimator.Estimator) estimator.Estimator {
	if engine == nil {
		return fallback
	}
	return &gpuEstimator{engine: engine, snapshot: snapshot, limiter: limiter, context: context, fastpath: fastpath, shared: shared,
		runner: runnerOf(snapshot), fallback: fallback}
}

// observeHeterogeneity restates estimator.observeBinpackingHeterogeneity (binpacking_estimator.go:371-396; unexported there, and
// package estimator does not change) on top of the exported metric (metrics/legacy_functions.go:245).
func observeHeterogeneity(pegs []estimator.PodEquivalenceGroup, tmpl *framework.NodeInfo) {
	var instanceType, cpuCount string
	if node := tmpl.Node(); node != nil {
		if node.Labels != nil {
			instanceType = node.Labels[apiv1.LabelInstanceTypeStable]
		}
		cpuCount = node.Status.Capacity.Cpu().String()
	}
	namespaces := map[string]bool{}
	for i := range pegs {
		if e := pegs[i].Exemplar(); e != nil {
			namespaces[e.Namespace] = true
		}
	}
	bucket := "11+" // (quantized: metric cardinality)
	if len(namespaces) <= 5 {
		bucket = strconv.Itoa(len(namespaces))
	} else if len(namespaces) <= 10 {
		bucket = "6-10"
	}
	metrics.ObserveBinpackingHeterogeneity(instanceType, cpuCount, bucket, len(pegs))
}

// Estimate implements estimator.Estimator (estimator.go:53-56) with the semantics of BinpackingNodeEstimator.Estimate
// (binpacking_estimator.go:102-161): node count, and the pods that fit in placement order.
func (g *gpuEstimator) Estimate(pegs []estimator.PodEquivalenceGroup, tmpl *framework.NodeInfo, ng cloudprovider.NodeGroup) (int, []*apiv1.Pod) {
	observeHeterogeneity(pegs, tmpl) // the metric stays on the Go side (binpacking_estimator.go:107)
	g.limiter.StartEstimation(pegs, ng, g.context)
	defer g.limiter.EndEstimation()
	maxNodes := g.limiter.MaxNodes()
	existing := nodeCount(g.snapshot)

	// ---- prefetch mode: the batch of this loop may hold the answer (prefetch.go) ----
	if g.shared != nil {
		if r, order, placed, ok := g.shared.lookup(ng, tmpl, pegs, maxNodes, existing); ok {
			if r.status == C.CASIM_NG_OK {
				// (a hit leaves the runner's lastIndex where the batch found it: every group of a batch starts from the same
				// loopLastIndex, INTEGRATION.md 1a; the per-call path below is the one that threads it from Estimate to Estimate)
				return int(r.node_count), prefixPods(pegs, order, placed)
			}
			return g.fallback.Estimate(pegs, tmpl, ng) // the batch delegated this group (CASIM_NG_UNSUPPORTED)
		}
	}

	// ---- per-call mode, first choice: the loop's tables are still there (a lookup missed: other PEG list, other limits) — no encoding ----
	if g.shared != nil {
		if n, order, placed, li, st, ok := g.shared.estimateOnLoopTables(ng, tmpl, pegs, maxNodes, existing, g.lastIndex(), g.fastpath); ok {
			if st != C.CASIM_NG_OK {
				return g.fallback.Estimate(pegs, tmpl, ng)
			}
			g.setLastIndex(li)
			return n, prefixPods(pegs, order, placed)
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
	scal := make([]C.int32_t, 6) // node_count, pods_scheduled, nodes_added, limiter_nodes, last_index_out, status
	sums := make([]C.int64_t, 2)
	order := make([]C.int32_t, n+1)
	placed := make([]C.int32_t, n+1)
	// a C struct that carries pointers into Go memory: every one of them pinned for the duration of the call (cgo pointer rules)
	var pin runtime.Pinner
	defer pin.Unpin()
	pin.Pin(&scal[0])
	pin.Pin(&sums[0])
	pin.Pin(&order[0])
	pin.Pin(&placed[0])
	res := C.casim_results{node_count: &scal[0], pods_scheduled: &scal[1], nodes_added: &scal[2], limiter_nodes: &scal[3],
		last_index_out: &scal[4], status: &scal[5], req_cpu_sum: &sums[0], req_mem_sum: &sums[1], order: &order[0], placed: &placed[0]}
	nodeCount, lastIndexOut, status := &scal[0], &scal[4], &scal[5]
	var opts C.casim_options
	if g.fastpath {
		opts.fastpath = 1
	}
	g.engine.mu.Lock()
	rc := C.casim_estimate_batch(g.engine.ctx, &pt, &gt, &opts, &res)
	g.engine.mu.Unlock()
	if rc != C.CASIM_OK || *status != C.CASIM_NG_OK {
		return g.fallback.Estimate(pegs, tmpl, ng) // fail closed: error, or a predicate outside the encoded subset
	}
	g.setLastIndex(int(*lastIndexOut)) // the runner's lastIndex persists across Estimates (plugin_runner.go:33-36,138)
	return int(*nodeCount), prefixPods(pegs, order[:n], placed[:n])
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
// runner's lastIndex (scheduling_opts.go:39-63) itself, per snapshot (runnerState), shared by every estimator built on that
// snapshot — with or without a prefetch cache.
func (g *gpuEstimator) lastIndex() int {
	g.runner.mu.Lock()
	defer g.runner.mu.Unlock()
	return g.runner.lastIndex
}
func (g *gpuEstimator) setLastIndex(v int) {
	g.runner.mu.Lock()
	g.runner.lastIndex = v
	g.runner.mu.Unlock()
}
