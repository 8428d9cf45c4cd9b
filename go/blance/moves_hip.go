// CalcPartitionMoves (moves.go:41-119) for every partition of two maps in one device call: the loop
// of OrchestrateMoves (orchestrate.go:273-287).  ADD to package blance next to plan_hip.go.

package blance

/*
#include <stdlib.h>
#include "blance_hip.h"
*/
import "C"

import (
	"runtime"
	"sort"
	"unsafe"
)

var moveOpNames = [...]string{"add", "del", "promote", "demote"} // BLANCE_OP_*

// CalcPartitionMovesBatch returns, per partition name, what CalcPartitionMoves(states, beg, end,
// favorMinNodes) returns; ok == false: no device, call CalcPartitionMoves per partition instead.
func CalcPartitionMovesBatch(
	states []string,
	begMap, endMap PartitionMap,
	favorMinNodes bool) (moves map[string][]NodeStateOp, ok bool) {
	ctx := hipContext()
	if !UseHIP || ctx == nil {
		return nil, false
	}
	names := make([]string, 0, len(begMap))
	seen := map[string]bool{}
	for n := range begMap {
		names, seen[n] = append(names, n), true
	}
	for n := range endMap {
		if !seen[n] {
			names = append(names, n)
		}
	}
	sort.Strings(names)
	M := len(states)
	sid := make(map[string]int, M)
	for i, s := range states {
		sid[s] = i
	}
	nodes := newInterner(1024)
	flatten := func(pm PartitionMap) (off, ids []int32) {
		off = append(off, 0)
		for _, n := range names {
			lists := make([][]string, M+1) // pseudo state M: keys that are not in `states`
			if p := pm[n]; p != nil {
				for st, lst := range p.NodesByState {
					if i, ok := sid[st]; ok {
						lists[i] = lst
					} else {
						lists[M] = append(lists[M], lst...)
					}
				}
			}
			for _, lst := range lists {
				for _, x := range lst {
					ids = append(ids, nodes.add(x))
				}
				off = append(off, int32(len(ids)))
			}
		}
		return
	}
	begOff, begNodes := flatten(begMap)
	endOff, endNodes := flatten(endMap)
	capOps := len(begNodes) + len(endNodes)
	opOff := make([]int32, len(names)+1)
	opNode := make([]int32, capOps+1)
	opState := make([]int32, capOps+1)
	opKind := make([]int32, capOps+1)

	var pin runtime.Pinner
	defer pin.Unpin()
	pb := (*C.blance_moves_problem)(C.calloc(1, C.size_t(unsafe.Sizeof(C.blance_moves_problem{}))))
	defer C.free(unsafe.Pointer(pb))
	res := (*C.blance_moves_result)(C.calloc(1, C.size_t(unsafe.Sizeof(C.blance_moves_result{}))))
	defer C.free(unsafe.Pointer(res))
	pb.n_parts = C.int32_t(len(names))
	pb.n_states = C.int32_t(M)
	pb.favor_min_nodes = b2i(favorMinNodes)
	pb.beg_off = i32(&pin, begOff)
	pb.beg_nodes = i32(&pin, begNodes)
	pb.end_off = i32(&pin, endOff)
	pb.end_nodes = i32(&pin, endNodes)
	res.op_off = i32(&pin, opOff)
	res.op_node = i32(&pin, opNode)
	res.op_state = i32(&pin, opState)
	res.op_kind = i32(&pin, opKind)
	res.capacity = C.int64_t(capOps)
	hipMu.Lock()
	st := C.blance_calc_moves(ctx, pb, res)
	hipMu.Unlock()
	if st != C.BLANCE_OK {
		return nil, false
	}
	moves = make(map[string][]NodeStateOp, len(names))
	for p, n := range names {
		var ops []NodeStateOp
		for j := opOff[p]; j < opOff[p+1]; j++ {
			state := ""
			if opState[j] >= 0 {
				state = states[opState[j]]
			}
			ops = append(ops, NodeStateOp{Node: nodes.names[opNode[j]], State: state, Op: moveOpNames[opKind[j]]})
		}
		moves[n] = ops
	}
	return moves, true
}
