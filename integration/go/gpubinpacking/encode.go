package gpubinpacking

/*
#include <stdlib.h>
#include "casim.h"
*/
import "C"

import (
	"runtime"

	apiv1 "k8s.io/api/core/v1"
	metav1 "k8s.io/apimachinery/pkg/apis/meta/v1"
	"k8s.io/autoscaler/cluster-autoscaler/estimator"
	"k8s.io/autoscaler/cluster-autoscaler/simulator/framework"
	podutils "k8s.io/autoscaler/cluster-autoscaler/utils/pod"
	schedutil "k8s.io/kubernetes/pkg/scheduler/util"
)

// session is one casim_encoder: pod specs and node templates in, flat tables out (casim_enc_tables).
type session struct {
	enc  *C.casim_encoder
	strs cstrings
	spec map[*apiv1.Pod]C.int32_t // exemplar pod -> pod-spec id of this session
	err  error                    // first value the encoder refused to record for a NODE (group): tables() returns it and the call falls back to the reference path
}

func newSession() *session {
	var o C.casim_encoder_options
	// three POSITIONAL lanes — cpu (milli), memory (bytes), ephemeral storage (bytes): the three fields framework.Resource keeps apart
	// (vendor/k8s.io/kubernetes/pkg/scheduler/framework/types.go:989-998).  Everything else the scheduler counts lives in
	// Resource.ScalarResources BY NAME and crosses the ABI by name (scalars below; casim_enc_pod_set_request /
	// casim_enc_group_set_allocatable, ABI 9): the encoder owns the lane numbers, this file cannot drop or mis-order one.
	o.n_res = 3
	return &session{enc: C.casim_enc_create(&o), spec: map[*apiv1.Pod]C.int32_t{}}
}

// scalars walks a ResourceList the way framework.Resource.Add does (types.go:1026-1048): cpu / memory / ephemeral-storage / pods have
// fields of their own, every other name counts iff schedutil.IsScalarResourceName says so (extended resources, hugepages-*,
// attachable-volumes-*, prefixed native names) and is then compared name by name in fitsRequest (noderesources/fit.go:731-763).
// NodeResourcesFit runs here with the default scheduler profile, as in the reference's framework handle: no ignored extended
// resources or resource groups, no DRA-backed extended resources (those pods are delegated: hasVolumesOrClaims).
func scalars(rl apiv1.ResourceList, visit func(name apiv1.ResourceName, value int64)) {
	for name, q := range rl {
		switch name {
		case apiv1.ResourceCPU, apiv1.ResourceMemory, apiv1.ResourceEphemeralStorage, apiv1.ResourcePods:
			continue
		}
		if schedutil.IsScalarResourceName(name) {
			visit(name, q.Value())
		}
	}
}

// request hands one named request of pod-spec id to the encoder and fails CLOSED: CASIM_ENC_DELEGATED (1: every lane taken) has already
// marked the pod CASIM_PEG_UNSUPPORTED; a NEGATIVE return (CASIM_ERR_INVALID: e.g. a negative Quantity.Value) means the request was NOT
// recorded, so the pod is marked here — its groups come back CASIM_NG_UNSUPPORTED and Estimate() runs the reference path.  A request is
// never silently ignored (ADVICE r5).
func (s *session) request(id C.int32_t, name apiv1.ResourceName, v int64) {
	if rc := C.casim_enc_pod_set_request(s.enc, id, s.strs.s(string(name)), C.int64_t(v)); rc < 0 {
		C.casim_enc_pod_mark_unsupported(s.enc, id, s.strs.s("request "+string(name)+" refused by the encoder"))
	}
}

func hasScalars(rl apiv1.ResourceList) bool {
	found := false
	scalars(rl, func(apiv1.ResourceName, int64) { found = true })
	return found
}

func (s *session) close() {
	if s.enc != nil {
		C.casim_enc_destroy(s.enc)
		s.enc = nil
	}
	s.strs.free()
}

func (s *session) selector(sel *metav1.LabelSelector, add func(key, op *C.char, vals **C.char, n C.int32_t)) {
	if sel == nil {
		return
	}
	for k, v := range sel.MatchLabels {
		add(s.strs.s(k), s.strs.s("In"), s.strs.arr([]string{v}), 1)
	}
	for _, r := range sel.MatchExpressions {
		add(s.strs.s(r.Key), s.strs.s(string(r.Operator)), s.strs.arr(r.Values), C.int32_t(len(r.Values)))
	}
}

// pod encodes what the Filter plugins read from the exemplar pod: NodeResourcesFit.PreFilter (computePodResourceRequest,
// vendor/k8s.io/kubernetes/pkg/scheduler/framework/plugins/noderesources/fit.go:321-331), TaintToleration, NodeAffinity /
// nodeSelector, NodePorts, InterPodAffinity, PodTopologySpread (SURVEY 8b "inputs the shim must extract").
func (s *session) pod(pod *apiv1.Pod) C.int32_t {
	if id, ok := s.spec[pod]; ok {
		return id
	}
	e, c := s.enc, &s.strs
	req := podutils.PodRequests(pod) // cluster-autoscaler/utils/pod/pod.go:88
	lanes := [C.CASIM_MAX_RES]C.int64_t{C.int64_t(req.Cpu().MilliValue()), C.int64_t(req.Memory().Value()), C.int64_t(req.StorageEphemeral().Value())}
	id := C.casim_enc_add_pod_spec(e, c.s(pod.Namespace), &lanes[0])
	s.spec[pod] = id
	// ScalarResources by name.  CASIM_ENC_DELEGATED (every lane taken): the encoder has marked the pod CASIM_PEG_UNSUPPORTED, its groups
	// come back CASIM_NG_UNSUPPORTED and Estimate() runs the reference path — a request is never silently ignored.
	scalars(req, func(name apiv1.ResourceName, v int64) { s.request(id, name, v) })
	for k, v := range pod.Labels {
		C.casim_enc_pod_add_label(e, id, c.s(k), c.s(v))
	}
	for _, t := range pod.Spec.Tolerations {
		C.casim_enc_pod_add_toleration(e, id, c.s(t.Key), c.s(string(t.Operator)), c.s(t.Value), c.s(string(t.Effect)))
	}
	for k, v := range pod.Spec.NodeSelector {
		C.casim_enc_pod_add_node_selector(e, id, c.s(k), c.s(v))
	}
	if len(pod.Spec.Containers) > 0 { // the fastpath chooser reads the FIRST container (binpacking_estimator.go:451-458)
		r := pod.Spec.Containers[0].Resources.Requests
		C.casim_enc_pod_set_fastpath_requests(e, id, C.double(r.Cpu().AsApproximateFloat64()), C.double(r.Memory().AsApproximateFloat64()))
	}
	s.podRest(pod, id)
	return id
}

// podRest: the fields casim_enc_add_pods (ABI 11) has no column for — host ports, pod (anti-)affinity, node-affinity terms, spread
// constraints, volumes / claims — on spec record id, after its namespace, requests, labels, tolerations and nodeSelector are in
// (pod by pod: pod() above; in one crossing: pegs() below).
func (s *session) podRest(pod *apiv1.Pod, id C.int32_t) {
	e, c := s.enc, &s.strs
	for _, p := range schedutil.GetHostPorts(pod) {
		C.casim_enc_pod_add_host_port(e, id, c.s(p.HostIP), c.s(string(p.Protocol)), C.int32_t(p.HostPort))
	}
	if a := pod.Spec.Affinity; a != nil {
		if a.PodAffinity != nil {
			for _, term := range a.PodAffinity.RequiredDuringSchedulingIgnoredDuringExecution {
				t := C.casim_enc_pod_add_affinity_term(e, id, c.s(term.TopologyKey), c.arr(term.Namespaces), C.int32_t(len(term.Namespaces)))
				if term.NamespaceSelector != nil { // the incoming pod's own term: resolved by casim_enc_finalize against the lister's namespaces
					C.casim_enc_aff_term_set_namespace_selector(e, id, t)
					s.selector(term.NamespaceSelector, func(k, op *C.char, v **C.char, n C.int32_t) {
						C.casim_enc_aff_term_add_namespace_requirement(e, id, t, k, op, v, n)
					})
				}
				s.selector(term.LabelSelector, func(k, op *C.char, v **C.char, n C.int32_t) { C.casim_enc_aff_term_add_requirement(e, id, t, k, op, v, n) })
			}
		}
		if a.PodAntiAffinity != nil {
			for _, term := range a.PodAntiAffinity.RequiredDuringSchedulingIgnoredDuringExecution {
				t := C.casim_enc_pod_add_anti_affinity_term(e, id, c.s(term.TopologyKey), c.arr(term.Namespaces), C.int32_t(len(term.Namespaces)))
				if term.NamespaceSelector != nil { // resolved by casim_enc_finalize against the namespaces fed with casim_enc_add_namespace
					C.casim_enc_term_set_namespace_selector(e, id, t)
					s.selector(term.NamespaceSelector, func(k, op *C.char, v **C.char, n C.int32_t) {
						C.casim_enc_term_add_namespace_requirement(e, id, t, k, op, v, n)
					})
				}
				s.selector(term.LabelSelector, func(k, op *C.char, v **C.char, n C.int32_t) { C.casim_enc_term_add_requirement(e, id, t, k, op, v, n) })
			}
		}
		if na := a.NodeAffinity; na != nil && na.RequiredDuringSchedulingIgnoredDuringExecution != nil {
			terms := na.RequiredDuringSchedulingIgnoredDuringExecution.NodeSelectorTerms
			if len(terms) == 0 { // a selector without terms matches nothing
				C.casim_enc_pod_add_node_affinity_term(e, id)
			}
			for _, term := range terms {
				t := C.casim_enc_pod_add_node_affinity_term(e, id)
				for _, r := range term.MatchExpressions {
					C.casim_enc_node_term_add_requirement(e, id, t, 0, c.s(r.Key), c.s(string(r.Operator)), c.arr(r.Values), C.int32_t(len(r.Values)))
				}
				for _, r := range term.MatchFields {
					C.casim_enc_node_term_add_requirement(e, id, t, 1, c.s(r.Key), c.s(string(r.Operator)), c.arr(r.Values), C.int32_t(len(r.Values)))
				}
			}
		}
	}
	for _, tc := range pod.Spec.TopologySpreadConstraints {
		if tc.WhenUnsatisfiable != apiv1.DoNotSchedule { // ScheduleAnyway only scores
			continue
		}
		minDomains := C.int32_t(0)
		if tc.MinDomains != nil {
			minDomains = C.int32_t(*tc.MinDomains)
		}
		ci := C.casim_enc_pod_add_spread_constraint(e, id, C.int32_t(tc.MaxSkew), c.s(tc.TopologyKey), minDomains)
		if tc.NodeTaintsPolicy != nil && *tc.NodeTaintsPolicy == apiv1.NodeInclusionPolicyHonor {
			C.casim_enc_spread_set_taints_policy(e, id, ci, 1)
		}
		if tc.NodeAffinityPolicy != nil && *tc.NodeAffinityPolicy == apiv1.NodeInclusionPolicyIgnore {
			C.casim_enc_spread_set_affinity_policy(e, id, ci, 0)
		}
		s.selector(tc.LabelSelector, func(k, op *C.char, v **C.char, n C.int32_t) { C.casim_enc_spread_add_requirement(e, id, ci, k, op, v, n) })
		for _, k := range tc.MatchLabelKeys { // the pod's own values join the selector (podtopologyspread/common.go:96-107)
			if v, ok := pod.Labels[k]; ok && tc.LabelSelector != nil {
				C.casim_enc_spread_add_requirement(e, id, ci, c.s(k), c.s("In"), c.arr([]string{v}), 1)
			}
		}
	}
	if hasVolumesOrClaims(pod) {
		C.casim_enc_pod_mark_unsupported(e, id, c.s("volumes / DRA"))
	}
}

func hasVolumesOrClaims(pod *apiv1.Pod) bool {
	for _, v := range pod.Spec.Volumes {
		if v.PersistentVolumeClaim != nil || v.Ephemeral != nil || v.CSI != nil {
			return true
		}
	}
	return len(pod.Spec.ResourceClaims) > 0
}

// peg adds one PodEquivalenceGroup: exemplar + size.
func (s *session) peg(g estimator.PodEquivalenceGroup) C.int32_t {
	return C.casim_enc_add_peg(s.enc, s.pod(g.Exemplar()), C.int32_t(len(g.Pods)))
}

// pegs adds the PodEquivalenceGroups of a loop in list order and returns their PEG ids.  What nearly every exemplar carries — namespace,
// the three positional requests, labels, tolerations, nodeSelector, the first container's requests — crosses the ABI ONCE
// (casim_enc_add_pods, ABI 11) as index columns over a string table interned here: one C string per DISTINCT key / value instead of one
// per use, 1 cgo crossing instead of ~9 per exemplar (C2 of the bench: 3 875 casim_enc_* calls pod by pod, 280 this way).  The encoder
// builds the same records as pod() + peg() would (tests/test_bulk_pods.py compares the tables); requests by name and the rarer
// fields follow per pod on the returned ids (scalars, podRest), in list order, so lanes are handed out as before.  An exemplar
// that already has a record in this session ends the run and takes peg().
func (s *session) pegs(groups []estimator.PodEquivalenceGroup) ([]C.int32_t, error) {
	ids := make([]C.int32_t, len(groups))
	var run []int
	inRun := map[*apiv1.Pod]bool{}
	var firstErr error
	flush := func() {
		if len(run) == 0 {
			return
		}
		n := len(run)
		strIdx := map[string]C.int32_t{}
		var strs []*C.char
		sid := func(x string) C.int32_t {
			if i, ok := strIdx[x]; ok {
				return i
			}
			i := C.int32_t(len(strs))
			strIdx[x] = i
			strs = append(strs, s.strs.s(x))
			return i
		}
		ns, cnt, out := make([]C.int32_t, n), make([]C.int32_t, n), make([]C.int32_t, n)
		req := make([]C.int64_t, 0, 3*n)
		fp := make([]C.double, 0, 2*n)
		loff, toff, soff := make([]C.int32_t, 1, n+1), make([]C.int32_t, 1, n+1), make([]C.int32_t, 1, n+1)
		var lk, lv, tk, to, tv, te, sk, sv []C.int32_t
		for i, gi := range run {
			pod := groups[gi].Exemplar()
			r := podutils.PodRequests(pod) // cluster-autoscaler/utils/pod/pod.go:88
			ns[i], cnt[i] = sid(pod.Namespace), C.int32_t(len(groups[gi].Pods))
			req = append(req, C.int64_t(r.Cpu().MilliValue()), C.int64_t(r.Memory().Value()), C.int64_t(r.StorageEphemeral().Value()))
			fc, fm := 0.0, 0.0
			if len(pod.Spec.Containers) > 0 { // the fastpath chooser reads the FIRST container (binpacking_estimator.go:451-458)
				cr := pod.Spec.Containers[0].Resources.Requests
				fc, fm = cr.Cpu().AsApproximateFloat64(), cr.Memory().AsApproximateFloat64()
			}
			fp = append(fp, C.double(fc), C.double(fm))
			for k, v := range pod.Labels {
				lk, lv = append(lk, sid(k)), append(lv, sid(v))
			}
			loff = append(loff, C.int32_t(len(lk)))
			for _, t := range pod.Spec.Tolerations {
				tk, to = append(tk, sid(t.Key)), append(to, sid(string(t.Operator)))
				tv, te = append(tv, sid(t.Value)), append(te, sid(string(t.Effect)))
			}
			toff = append(toff, C.int32_t(len(tk)))
			for k, v := range pod.Spec.NodeSelector {
				sk, sv = append(sk, sid(k)), append(sv, sid(v))
			}
			soff = append(soff, C.int32_t(len(sk)))
		}
		// the struct carries pointers into Go memory: every column is pinned for the call (cgo pointer rules)
		var pin runtime.Pinner
		defer pin.Unpin()
		col := func(v []C.int32_t) *C.int32_t {
			if len(v) == 0 {
				return nil
			}
			pin.Pin(&v[0])
			return &v[0]
		}
		var pc C.casim_pod_columns
		pc.n_pods, pc.n_strings = C.int32_t(n), C.int32_t(len(strs))
		pin.Pin(&strs[0])
		pc.strings = &strs[0]
		pin.Pin(&req[0])
		pin.Pin(&fp[0])
		pc.ns, pc.req, pc.fastpath_req, pc.peg_count = col(ns), &req[0], &fp[0], col(cnt)
		pc.label_off, pc.label_key, pc.label_val = col(loff), col(lk), col(lv)
		pc.tol_off, pc.tol_key, pc.tol_op, pc.tol_value, pc.tol_effect = col(toff), col(tk), col(to), col(tv), col(te)
		pc.sel_off, pc.sel_key, pc.sel_val = col(soff), col(sk), col(sv)
		first := C.casim_enc_add_pods(s.enc, &pc, &out[0])
		if first < 0 { // nothing was added (casim.h): the caller falls back to the reference path for this loop
			if firstErr == nil {
				firstErr = rcErr(first, "casim_enc_add_pods")
			}
		} else {
			for i, gi := range run {
				pod, id := groups[gi].Exemplar(), first+C.int32_t(i)
				s.spec[pod] = id
				// ScalarResources by name, in list order (CASIM_ENC_DELEGATED marks the pod: see pod())
				scalars(podutils.PodRequests(pod), func(name apiv1.ResourceName, v int64) { s.request(id, name, v) })
				s.podRest(pod, id)
				ids[gi] = out[i]
			}
		}
		run = run[:0]
		for k := range inRun {
			delete(inRun, k)
		}
	}
	for gi, g := range groups {
		pod := g.Exemplar()
		if _, seen := s.spec[pod]; seen || inRun[pod] {
			flush()
			ids[gi] = s.peg(g)
			continue
		}
		run = append(run, gi)
		inRun[pod] = true
	}
	flush()
	return ids, firstErr
}

// group adds one node group: the template the estimator clones for every simulated node (SanitizedNodeInfo,
// cluster-autoscaler/simulator/node_info_utils.go:93-137) with the DaemonSet pods preloaded on it, the limiter's answer,
// the snapshot's node count E and the runner's lastIndex.  pegs == nil: SchedulablePodGroups is derived on the device.
func (s *session) group(tmpl *framework.NodeInfo, maxNodes, existing, lastIndex int, pegs []C.int32_t) C.int32_t {
	node, c := tmpl.Node(), &s.strs
	al := node.Status.Allocatable
	lanes := [C.CASIM_MAX_RES]C.int64_t{C.int64_t(al.Cpu().MilliValue()), C.int64_t(al.Memory().Value()), C.int64_t(al.StorageEphemeral().Value())}
	unsched := C.int32_t(0)
	if node.Spec.Unschedulable {
		unsched = 1
	}
	g := C.casim_enc_add_group(s.enc, c.s(node.Name), &lanes[0], C.int32_t(al.Pods().Value()),
		C.int64_t(node.Status.Capacity.Cpu().MilliValue()), C.int64_t(node.Status.Capacity.Memory().Value()), unsched)
	// Allocatable.ScalarResources by name (NodeInfo.SetNode -> NewResource(node.Status.Allocatable), types.go:1003-1011)
	// (the encoder opens a lane for a name only when some pod asks for a non-zero amount of it: hugepages-*: 0 and attachable-volumes-* of
	// real nodes widen no table; a negative return means the value was NOT recorded — the session fails closed)
	scalars(al, func(name apiv1.ResourceName, v int64) {
		if rc := C.casim_enc_group_set_allocatable(s.enc, g, c.s(string(name)), C.int64_t(v)); rc < 0 && s.err == nil {
			s.err = rcErr(rc, "casim_enc_group_set_allocatable("+string(name)+")")
		}
	})
	for k, v := range node.Labels {
		C.casim_enc_group_add_label(s.enc, g, c.s(k), c.s(v))
	}
	for _, t := range node.Spec.Taints {
		C.casim_enc_group_add_taint(s.enc, g, c.s(t.Key), c.s(t.Value), c.s(string(t.Effect)))
	}
	C.casim_enc_group_set_fastpath_capacity(s.enc, g, C.double(node.Status.Capacity.Cpu().AsApproximateFloat64()), C.double(node.Status.Capacity.Memory().AsApproximateFloat64()))
	C.casim_enc_group_set_limits(s.enc, g, C.int32_t(maxNodes), C.int32_t(existing), C.int32_t(lastIndex))
	for _, pi := range tmpl.Pods() {
		C.casim_enc_group_add_preloaded_pod(s.enc, g, s.pod(pi.Pod))
	}
	if pegs != nil {
		var p *C.int32_t
		if len(pegs) > 0 {
			p = &pegs[0]
		}
		C.casim_enc_group_set_pegs(s.enc, g, p, C.int32_t(len(pegs)))
	}
	return g
}

func (s *session) tables() (pegs C.casim_pegs, groups C.casim_groups, err error) {
	if err = s.err; err != nil {
		return
	}
	if err = rcErr(C.casim_enc_finalize(s.enc), "casim_enc_finalize"); err != nil {
		return
	}
	err = rcErr(C.casim_enc_tables(s.enc, &pegs, &groups), "casim_enc_tables")
	return
}

// runningPods hands the running pods of many nodes to the encoder in ONE cgo crossing (casim_enc_add_running_pods; per-node
// consumers — filter-out-schedulable, the removal simulation — describe 10^5 running pods per loop).  Pods that only carry a
// namespace, labels and requests go into flat arrays with an interned string table (no C string per label); a pod with
// tolerations, selectors, host ports or (anti-)affinity terms ends the batch collected so far and takes the per-pod calls, so
// that spec ids follow the order of the pods.  groups[i] is the encoder's group id of nodes[i].
func (s *session) runningPods(nodes []*framework.NodeInfo, groups []C.int32_t) error {
	var grp, ns, off, lk, lv []C.int32_t
	var req []C.int64_t
	off = append(off, 0)
	strIdx := map[string]C.int32_t{}
	var strs []*C.char
	sid := func(x string) C.int32_t {
		if i, ok := strIdx[x]; ok {
			return i
		}
		i := C.int32_t(len(strs))
		strIdx[x] = i
		strs = append(strs, s.strs.s(x))
		return i
	}
	var firstErr error
	flush := func() {
		if len(grp) == 0 {
			return
		}
		var pk, pv *C.int32_t
		if len(lk) > 0 {
			pk, pv = &lk[0], &lv[0]
		}
		// the call adds NOTHING on a bad index (casim.h): a dropped batch would leave nodes emptier than they are, so the
		// first error ends the session — the caller falls back to the Go path for this loop
		if rc := C.casim_enc_add_running_pods(s.enc, C.int32_t(len(grp)), &grp[0], &ns[0], &req[0], &off[0], pk, pv, &strs[0], C.int32_t(len(strs))); rc < 0 && firstErr == nil {
			firstErr = rcErr(rc, "casim_enc_add_running_pods")
		}
		grp, ns, req, lk, lv, off = grp[:0], ns[:0], req[:0], lk[:0], lv[:0], append(off[:0], 0)
	}
	plain := func(p *apiv1.Pod) bool {
		sp := &p.Spec
		if len(sp.Tolerations) > 0 || len(sp.NodeSelector) > 0 || sp.Affinity != nil || len(sp.TopologySpreadConstraints) > 0 || hasVolumesOrClaims(p) {
			return false
		}
		for i := range sp.Containers {
			for _, cp := range sp.Containers[i].Ports {
				if cp.HostPort > 0 {
					return false
				}
			}
		}
		return true
	}
	for i, ni := range nodes {
		for _, pi := range ni.Pods() {
			p := pi.Pod
			r := podutils.PodRequests(p)
			// (the flat arrays carry the three positional lanes: a pod with scalar / extended requests takes the per-pod calls, which name them)
			if _, seen := s.spec[p]; seen || !plain(p) || hasScalars(r) {
				flush()
				C.casim_enc_group_add_preloaded_pod(s.enc, groups[i], s.pod(p))
				continue
			}
			grp = append(grp, groups[i])
			ns = append(ns, sid(p.Namespace))
			req = append(req, C.int64_t(r.Cpu().MilliValue()), C.int64_t(r.Memory().Value()), C.int64_t(r.StorageEphemeral().Value()))
			for k, v := range p.Labels { // (any order: the encoder keeps a pod's labels sorted by key)
				lk = append(lk, sid(k))
				lv = append(lv, sid(v))
			}
			off = append(off, C.int32_t(len(lk)))
		}
	}
	flush()
	return firstErr
}
