// The cgo side of the boundary: planNextMapEx (plan.go:23-58) on an MI355X through
// include/blance_hip.h.  ADD this file (with intern.go and moves_hip.go) to package blance and
// route api.go:155 through planNextMapHip -- see go/blance/README.md for the two-line patch.
//
// Build: CGO_CFLAGS=-I<repo>/include CGO_LDFLAGS="-L<repo>/blance_amd/lib -lblance_hip" go build
// (Go >= 1.21 for runtime.Pinner).  Cannot be compiled in this repository's build image (no Go
// toolchain); its C++ twin blance_amd/csrc/host/blance_api.cpp is what the tests exercise.

package blance

/*
#include <stdlib.h>
#include "blance_hip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"reflect"
	"runtime"
	"strconv"
	"sync"
	"unsafe"
)

// UseHIP switches the device path on (default: on when a context can be created).
var UseHIP = true

// CbgtNodeScoreBooster is the one booster known in the wild (couchbase/cbgt, restated in
// control_test.go:19-26).  Assign THIS function to NodeScoreBooster to keep the device path:
// an arbitrary Go callback cannot run on the GPU.
func CbgtNodeScoreBooster(w int, stickiness float64) float64 {
	score := float64(-w)
	if score < stickiness {
		score = stickiness
	}
	return score
}

var (
	hipOnce sync.Once
	hipCtx  *C.blance_ctx
	hipMu   sync.Mutex // the library serialises calls per context as well
)

// The API has no error channel (api.go:147-154): a call the device does not answer -- switched off, a custom sorter or
// booster, no device, an input outside the envelope (INTEGRATION.md section 5), a device failure -- is answered by
// plan.go's planner, silently as far as the caller's result goes.  HipStats and OnHipFallback say that it happened.
type HipCounters struct {
	DevicePlans uint64 // calls answered by the device
	Fallbacks   uint64 // calls handed to the CPU planner
	LastReason  string // why the last of those was (the refusal's text, or blance_last_error's)
}

var (
	hipStatsMu sync.Mutex
	hipStats   HipCounters
	// OnHipFallback, when set, is called (outside every lock) with the reason of each call the CPU planner answers.
	OnHipFallback func(reason string)
)

// HipStats returns the counters so far.
func HipStats() HipCounters {
	hipStatsMu.Lock()
	defer hipStatsMu.Unlock()
	return hipStats
}

func hipFellBack(reason string) {
	hipStatsMu.Lock()
	hipStats.Fallbacks++
	hipStats.LastReason = reason
	cb := OnHipFallback
	hipStatsMu.Unlock()
	if cb != nil {
		cb(reason)
	}
}

func hipAnswered() {
	hipStatsMu.Lock()
	hipStats.DevicePlans++
	hipStatsMu.Unlock()
}

func hipContext() *C.blance_ctx {
	hipOnce.Do(func() {
		var opt C.blance_options
		if v, err := strconv.Atoi(os.Getenv("BLANCE_HIP_DEVICE")); err == nil {
			opt.device_id = C.int32_t(v)
		}
		var ctx *C.blance_ctx
		if C.blance_ctx_create(&opt, &ctx) == C.BLANCE_OK {
			hipCtx = ctx
		}
	})
	return hipCtx
}

func funcPointer(f interface{}) uintptr {
	if f == nil {
		return 0
	}
	v := reflect.ValueOf(f)
	if v.Kind() != reflect.Func || v.IsNil() {
		return 0
	}
	return v.Pointer()
}

// boosterKindForDevice: which built-in booster the package globals ask for, or an error when the
// device cannot honour them (plan.go:580 CustomNodeSorter, plan.go:693 NodeScoreBooster).
func boosterKindForDevice() (int, error) {
	if funcPointer(CustomNodeSorter) != funcPointer(defaultNodeSorter) {
		return 0, unsupported("CustomNodeSorter is not the default sorter")
	}
	if NodeScoreBooster == nil {
		return int(C.BLANCE_BOOSTER_NONE), nil
	}
	if funcPointer(NodeScoreBooster) == funcPointer(CbgtNodeScoreBooster) {
		return int(C.BLANCE_BOOSTER_CBGT), nil
	}
	return 0, unsupported("NodeScoreBooster is an arbitrary Go callback")
}

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

func i32(p *runtime.Pinner, s []int32) *C.int32_t {
	if len(s) == 0 {
		s = []int32{0}
	}
	p.Pin(&s[0])
	return (*C.int32_t)(unsafe.Pointer(&s[0]))
}

func u8(p *runtime.Pinner, s []uint8) *C.uint8_t {
	if len(s) == 0 {
		s = []uint8{0}
	}
	p.Pin(&s[0])
	return (*C.uint8_t)(unsafe.Pointer(&s[0]))
}

// planNextMapHip is planNextMapEx on the device.  handled == false: the input is outside the
// device envelope or no device is available -- the caller runs plan.go's planNextMapEx.
func planNextMapHip(
	prevMap PartitionMap,
	partitionsToAssign PartitionMap,
	nodesAll []string,
	nodesToRemove []string,
	nodesToAdd []string,
	model PartitionModel,
	options PlanNextMapOptions) (nextMap PartitionMap, warnings map[string][]string, handled bool) {
	if !UseHIP {
		hipFellBack("UseHIP is false")
		return nil, nil, false
	}
	kind, err := boosterKindForDevice()
	if err != nil {
		hipFellBack(err.Error())
		return nil, nil, false
	}
	ctx := hipContext()
	if ctx == nil {
		hipFellBack("no device context (blance_ctx_create failed)")
		return nil, nil, false
	}
	f, err := internProblem(prevMap, partitionsToAssign, nodesAll, nodesToRemove, nodesToAdd, model, options, kind)
	if err != nil {
		hipFellBack(err.Error())
		return nil, nil, false
	}

	// blance_problem lives in C memory; the Go arrays it points to are pinned for the call
	// (cgo pointer rules), nothing is retained by the library after it returns.
	var pin runtime.Pinner
	defer pin.Unpin()
	pb := (*C.blance_problem)(C.calloc(1, C.size_t(unsafe.Sizeof(C.blance_problem{}))))
	defer C.free(unsafe.Pointer(pb))
	pb.n_nodes = C.int32_t(f.nNodes)
	pb.n_nodes_ext = C.int32_t(f.nNodesExt)
	pb.n_states = C.int32_t(f.nStates)
	pb.n_parts = C.int32_t(f.nParts)
	pb.n_prev = C.int32_t(f.nPrev)
	pb.n_loads = C.int32_t(len(f.loadState))
	pb.n_rules = C.int32_t(len(f.ruleInc))
	pb.n_vertices = C.int32_t(f.nVertices)
	pb.max_iterations = C.int32_t(f.maxIterations)
	pb.partition_weights_nil = b2i(f.partitionWeightsNil)
	pb.nodes_to_add_nil = b2i(f.nodesToAddNil)
	pb.hierarchy_rules_nil = b2i(f.hierarchyRulesNil)
	pb.booster_kind = C.int32_t(f.boosterKind)
	pb.top_state = C.int32_t(f.topState)
	pb.state_priority = i32(&pin, f.statePriority)
	pb.state_constraints = i32(&pin, f.stateConstraints)
	pb.state_stickiness = i32(&pin, f.stateStickiness)
	pb.state_has_stickiness = u8(&pin, f.stateHasStickiness)
	pb.node_removed = u8(&pin, f.nodeRemoved)
	pb.node_added = u8(&pin, f.nodeAdded)
	pb.node_weight = i32(&pin, f.nodeWeight)
	pb.node_has_weight = u8(&pin, f.nodeHasWeight)
	pb.part_order = i32(&pin, f.partOrder)
	pb.part_weight = i32(&pin, f.partWeight)
	pb.part_has_weight = u8(&pin, f.partHasWeight)
	pb.part_in_prev = u8(&pin, f.partInPrev)
	pb.part_prev_never_equal = u8(&pin, f.partPrevNeverEqual)
	pb.assign_off = i32(&pin, f.assignOff)
	pb.assign_nodes = i32(&pin, f.assignNodes)
	pb.assign_kind = u8(&pin, f.assignKind)
	pb.prev_off = i32(&pin, f.prevOff)
	pb.prev_nodes = i32(&pin, f.prevNodes)
	pb.prev_kind = u8(&pin, f.prevKind)
	pb.load_state = i32(&pin, f.loadState)
	pb.load_node = i32(&pin, f.loadNode)
	pb.load_weight = i32(&pin, f.loadWeight)
	pb.load_first_sweep_only = u8(&pin, f.loadFirstSweepOnly)
	pb.rule_off = i32(&pin, f.ruleOff)
	pb.rule_inc = i32(&pin, f.ruleInc)
	pb.rule_exc = i32(&pin, f.ruleExc)
	pb.vertex_empty = C.int32_t(f.vertexEmpty)
	pb.vertex_parent = i32(&pin, f.vertexParent)
	pb.vertex_leaf_lo = i32(&pin, f.vertexLeafLo)
	pb.vertex_leaf_hi = i32(&pin, f.vertexLeafHi)
	pb.node_leaf_pos = i32(&pin, f.nodeLeafPos)

	// No blance_validate / blance_result_capacity in front of the call (ABI 5): blance_plan makes the same checks itself,
	// the O(P) ones on the device, and returns the same status; the capacity is bounded here without another walk:
	// sum of max(k, len) <= sum of len + P * sum of k.  The arrays above are ordinary Go slices (pageable): the library
	// stages them through its own page-locked buffer with a few threads; a caller that wants the DMA to read them where
	// they lie allocates them with C.blance_host_alloc (include/blance_hip.h) -- the ownership rule stays "caller owns".
	M, P := f.nStates, f.nParts
	PM := P * M
	capNodes := int64(len(f.assignNodes))
	for m := 0; m < M; m++ {
		if k := int64(f.stateConstraints[m]); k > 0 {
			capNodes += int64(P) * k
		}
	}
	outOff := make([]int32, PM+1)
	outNodes := make([]int32, capNodes+1)
	outKind := make([]uint8, PM+1)
	warnPart := make([]int32, PM+1)
	warnState := make([]int32, PM+1)
	res := (*C.blance_result)(C.calloc(1, C.size_t(unsafe.Sizeof(C.blance_result{}))))
	defer C.free(unsafe.Pointer(res))
	res.out_off = i32(&pin, outOff)
	res.out_nodes = i32(&pin, outNodes)
	res.out_kind = u8(&pin, outKind)
	res.out_capacity = C.int64_t(capNodes)
	res.warn_part = i32(&pin, warnPart)
	res.warn_state = i32(&pin, warnState)
	res.warn_capacity = C.int64_t(PM)

	hipMu.Lock()
	runtime.LockOSThread() // blance_last_error is per OS thread: the two calls must not be split across threads
	st := C.blance_plan(ctx, pb, res)
	why := ""
	if st != C.BLANCE_OK {
		why = fmt.Sprintf("blance_plan status %d: %s", int(st), C.GoString(C.blance_last_error()))
	}
	runtime.UnlockOSThread()
	hipMu.Unlock()
	if st != C.BLANCE_OK {
		// BLANCE_ERR_UNSUPPORTED: outside the envelope; anything else: the device failed -- the API
		// has no error channel (api.go:147-154), so the CPU planner answers in both cases
		hipFellBack(why)
		return nil, nil, false
	}
	hipAnswered()
	if int(res.iterations) == 0 { // MaxIterationsPerPlan <= 0: planNextMapEx returns (nil, nil)
		return nil, nil, true
	}

	// ---- ids back to strings: fresh *Partition objects, every state key of the input carried through.
	// Three allocations carry them instead of five per partition: one []Partition, one []string holding the node names
	// of every list back to back (a list is a full slice expression of it -- cap == len --, so an append by the caller
	// reallocates instead of running into the next list), and the result map sized up front.  The NodesByState map of a
	// partition stays one make each (Go cannot carve maps out of a block).  When the call converged in sweep n > 1 the
	// input maps get a SECOND set of objects (below); it comes out of the same two blocks.
	stores := int(res.iterations) > 1 || res.converged == 0
	copies := 1
	if stores && res.converged != 0 {
		copies = 2
	}
	total := int(outOff[PM])
	parts := make([]Partition, copies*P)
	names := make([]string, copies*total)
	fill := func(part *Partition, p int, base int) {
		part.Name = f.partNames[p]
		part.NodesByState = make(map[string][]string, M)
		for m := 0; m < M; m++ {
			i := p*M + m
			switch outKind[i] {
			case listAbsent:
			case listNil:
				part.NodesByState[f.stateNames[m]] = nil
			default:
				lo, hi := base+int(outOff[i]), base+int(outOff[i+1])
				for j := lo; j < hi; j++ {
					names[j] = f.nodeNames[outNodes[j-base]]
				}
				part.NodesByState[f.stateNames[m]] = names[lo:hi:hi] // non-nil also when empty
			}
		}
	}
	nextMap = make(PartitionMap, P)
	for p := 0; p < P; p++ {
		fill(&parts[p], p, 0)
		nextMap[parts[p].Name] = &parts[p]
	}
	warnings = map[string][]string{}
	for i := 0; i < int(res.n_warnings); i++ { // plan.go:231-234, the reference's own text
		name, state := f.partNames[warnPart[i]], f.stateNames[warnState[i]]
		warnings[name] = append(warnings[name],
			fmt.Sprintf("could not meet constraints: %d, stateName: %s, partitionName: %s",
				f.stateConstraints[warnState[i]], state, name))
	}
	// plan.go:49-52: every sweep that did not converge stores its partitions into BOTH caller maps;
	// the last such store carries the final content (a converging last sweep changes nothing).
	// When the call converged (in sweep n > 1) the stored objects are sweep n-1's: equal in content to the
	// returned ones, but not the same objects (plan.go:334-343 makes fresh ones every sweep) -- a caller that
	// edits nextMap[p] afterwards must not edit prevMap[p].  At the iteration cap the returned objects ARE
	// the stored ones.  One clone per partition, shared by both maps as in the reference.
	if stores {
		for p := 0; p < P; p++ {
			stored := &parts[p]
			if copies == 2 {
				stored = &parts[P+p]
				fill(stored, p, total)
			}
			prevMap[stored.Name] = stored
			partitionsToAssign[stored.Name] = stored
		}
	}
	return nextMap, warnings, true
}
