// Table test for the device path, for a machine that has Go, an MI355X and libblance_hip.so: replays the
// reference's own golden cases as transcribed in this repository (tests/golden/planner_cases.json, produced by
// tools/extract_golden.py from plan_test.go / control_test.go) through planNextMapHip and through plan.go's
// planNextMapEx, and requires identical maps, warnings and caller-visible mutations.
//
//	BLANCE_GOLDEN=/path/to/tests/golden/planner_cases.json go test -run TestHipMatchesPlanGo
//
// Not run in this repository's build image (no Go toolchain); the same cases run against the same C ABI through
// the C++ twin of the shim (tests/test_host_cpp.py).

package blance

import (
	"encoding/json"
	"os"
	"reflect"
	"testing"
)

type goldenCase struct {
	Source                string                     `json:"source"`
	Ignored               bool                       `json:"ignored"`
	Aliased               bool                       `json:"aliased"`
	PrevMap               PartitionMap               `json:"prevMap"`
	PartitionsToAssign    PartitionMap               `json:"partitionsToAssign"`
	NodesAll              []string                   `json:"nodesAll"`
	NodesToRemove         []string                   `json:"nodesToRemove"`
	NodesToAdd            []string                   `json:"nodesToAdd"`
	Model                 PartitionModel             `json:"model"`
	ModelStateConstraints map[string]int             `json:"modelStateConstraints"`
	PartitionWeights      map[string]int             `json:"partitionWeights"`
	StateStickiness       map[string]int             `json:"stateStickiness"`
	NodeWeights           map[string]int             `json:"nodeWeights"`
	NodeHierarchy         map[string]string          `json:"nodeHierarchy"`
	HierarchyRules        HierarchyRules             `json:"hierarchyRules"`
	Booster               string                     `json:"booster"`
	Exp                   PartitionMap               `json:"exp"`
}

func deepCopyMap(m PartitionMap) PartitionMap {
	if m == nil {
		return nil
	}
	out := PartitionMap{}
	for k, p := range m {
		out[k] = &Partition{Name: p.Name, NodesByState: copyNodesByState(p.NodesByState)}
	}
	return out
}

func TestHipMatchesPlanGo(t *testing.T) {
	path := os.Getenv("BLANCE_GOLDEN")
	if path == "" {
		t.Skip("BLANCE_GOLDEN not set")
	}
	raw, err := os.ReadFile(path)
	if err != nil {
		t.Fatal(err)
	}
	var doc struct {
		Cases []goldenCase `json:"cases"`
	}
	if err := json.Unmarshal(raw, &doc); err != nil {
		t.Fatal(err)
	}
	for _, c := range doc.Cases {
		if c.Ignored {
			continue
		}
		NodeScoreBooster = nil
		if c.Booster == "cbgt" {
			NodeScoreBooster = CbgtNodeScoreBooster
		}
		opts := PlanNextMapOptions{
			ModelStateConstraints: c.ModelStateConstraints, PartitionWeights: c.PartitionWeights,
			StateStickiness: c.StateStickiness, NodeWeights: c.NodeWeights,
			NodeHierarchy: c.NodeHierarchy, HierarchyRules: c.HierarchyRules,
		}
		run := func(hip bool) (PartitionMap, map[string][]string, PartitionMap, PartitionMap) {
			prev := deepCopyMap(c.PrevMap)
			assign := deepCopyMap(c.PartitionsToAssign)
			if c.Aliased { // the Vis tests pass ONE map as both arguments (plan_test.go:1716-1718)
				assign = prev
			}
			if hip {
				next, warn, handled := planNextMapHip(prev, assign, c.NodesAll, c.NodesToRemove, c.NodesToAdd, c.Model, opts)
				if !handled {
					t.Fatalf("%s: not handled by the device path", c.Source)
				}
				return next, warn, prev, assign
			}
			next, warn := planNextMapEx(prev, assign, c.NodesAll, c.NodesToRemove, c.NodesToAdd, c.Model, opts)
			return next, warn, prev, assign
		}
		gn, gw, gp, ga := run(true)
		wn, ww, wp, wa := run(false)
		if !reflect.DeepEqual(gn, wn) || !reflect.DeepEqual(gn, c.Exp) {
			t.Errorf("%s: nextMap differs", c.Source)
		}
		if len(gw) != len(ww) || (len(gw) > 0 && !reflect.DeepEqual(gw, ww)) {
			t.Errorf("%s: warnings differ", c.Source)
		}
		if !reflect.DeepEqual(gp, wp) || !reflect.DeepEqual(ga, wa) {
			t.Errorf("%s: the caller's maps were mutated differently (plan.go:49-52)", c.Source)
		}
	}
	NodeScoreBooster = nil
}

// A call the device does not answer is reported: counters and the OnHipFallback hook (the API itself has no error channel).
func TestHipFallbackIsReported(t *testing.T) {
	var got []string
	OnHipFallback = func(reason string) { got = append(got, reason) }
	defer func() { OnHipFallback = nil; NodeScoreBooster = nil }()
	before := HipStats()
	NodeScoreBooster = func(w int, s float64) float64 { return 0 } // an arbitrary callback: not for the device
	model := PartitionModel{"primary": &PartitionModelState{Priority: 0, Constraints: 1}}
	parts := PartitionMap{"0": &Partition{Name: "0", NodesByState: map[string][]string{}}}
	_, _, handled := planNextMapHip(PartitionMap{}, parts, []string{"a", "b"}, nil, []string{"a", "b"}, model, PlanNextMapOptions{})
	after := HipStats()
	if handled || after.Fallbacks != before.Fallbacks+1 || len(got) != 1 || got[0] == "" || after.LastReason != got[0] {
		t.Fatalf("fallback not reported: handled %v, %+v -> %+v, hook %q", handled, before, after, got)
	}
}
