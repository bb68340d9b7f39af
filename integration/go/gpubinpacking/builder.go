package gpubinpacking

import (
	"time"

	"k8s.io/autoscaler/cluster-autoscaler/cloudprovider"
	"k8s.io/autoscaler/cluster-autoscaler/estimator"
	"k8s.io/autoscaler/cluster-autoscaler/simulator/clustersnapshot"
)

// deviceLimiter is the reference's thresholdBasedEstimationLimiter plus the one number the device needs from it.  The reference
// keeps maxNodes private (threshold_based_limiter.go:27-32); the thresholds and their interface are public (threshold.go), so the
// shim folds them with the reference's own rule (getMinLimit, :45-54) next to the wrapped limiter — which still sees every
// StartEstimation / EndEstimation exactly as today.  No file of package estimator changes.
type deviceLimiter struct {
	estimator.EstimationLimiter
	thresholds []estimator.Threshold
	maxNodes   int
}

func newDeviceLimiter(thresholds []estimator.Threshold) *deviceLimiter {
	return &deviceLimiter{EstimationLimiter: estimator.NewThresholdBasedEstimationLimiter(thresholds), thresholds: thresholds}
}

func minLimit(base, target int) int { // getMinLimit, threshold_based_limiter.go:45-54
	if base < 0 || target < 0 {
		return -1
	}
	if (base == 0 || base > target) && target > 0 {
		return target
	}
	return base
}

func (l *deviceLimiter) StartEstimation(pegs []estimator.PodEquivalenceGroup, ng cloudprovider.NodeGroup, ctx estimator.EstimationContext) {
	l.EstimationLimiter.StartEstimation(pegs, ng, ctx)
	l.maxNodes = 0
	for _, t := range l.thresholds {
		l.maxNodes = minLimit(l.maxNodes, t.NodeLimit(ng, ctx))
		if t.DurationLimit(ng, ctx) < time.Duration(0) { // a negative duration limit forbids the estimation like a negative node limit
			l.maxNodes = -1
		}
	}
}

func (l *deviceLimiter) MaxNodes() int { return l.maxNodes }

// BuilderOptions: what a host may set besides the reference's own arguments.
type BuilderOptions struct {
	Routing             Routing  // which per-call Estimates go to the device (estimator.go); zero value = DefaultRouting
	AnalyserOnReference bool     // an EstimationAnalyserFunc that reads the PODS of simulated nodes: every Estimate runs the reference estimator
	Runners             *Runners // nil = a registry of its own
}

// NewEstimatorBuilder is what core/autoscaler.go assigns to AutoscalerOptions.EstimatorBuilder when --estimator=gpu-binpacking
// (autoscaler_go.patch): the same arguments estimator.NewEstimatorBuilder takes, with the thresholds instead of the limiter built
// from them.  engine == nil (no MI355X, ABI mismatch): every estimator it builds IS the reference's BinpackingNodeEstimator.
func NewEstimatorBuilder(engine *Engine, shared *Shared, thresholds []estimator.Threshold, orderer estimator.EstimationPodOrderer,
	analyser estimator.EstimationAnalyserFunc, fastpath bool, bo BuilderOptions) estimator.EstimatorBuilder {
	limiter := newDeviceLimiter(thresholds)
	runners := bo.Runners
	if runners == nil {
		runners = NewRunners()
	}
	routing := bo.Routing
	if routing == (Routing{}) {
		routing = DefaultRouting
	}
	if shared != nil {
		shared.limiter = limiter
		shared.runners = runners
	}
	return func(snapshot clustersnapshot.ClusterSnapshot, ctx estimator.EstimationContext) estimator.Estimator {
		fallback := estimator.NewBinpackingNodeEstimator(snapshot, limiter, orderer, ctx, analyser, fastpath)
		if analyser != nil && bo.AnalyserOnReference {
			return fallback
		}
		// with an analyser the estimator takes the per-call path with casim_options.node_pods and calls it (estimator.go: analyse); the
		// fallback it falls back to calls it by itself
		return New(engine, snapshot, limiter, ctx, fastpath, shared, runners, analyser, routing, fallback)
	}
}
