import json, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in sys.argv[1:]:
    try:
        t = open(os.path.join(root, 'gpurun_out', 'sh%s.log' % b)).read()
        d = json.loads(t[t.index('{'):])
        print(b, {k: (v['first_ms'], v['diag']['shared_blocks'], v['diag']['shared_tasks']) for k, v in d.items()})
    except Exception as e:
        print(b, 'unreadable:', repr(e))
