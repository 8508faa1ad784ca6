import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
dd = d['dropin']
print('step', d['ms_per_step'], {k: dd.get(k) for k in ('ms_per_step', 'ms_per_step_without_loss_item', 'ms_per_step_median', 'ms_slowest_step', 'cgroup_cpu_throttled_during_steps', 'python_gc_collections_during_steps')},
      'redirected', {k: dd['with_optim_import_redirected'].get(k) for k in ('ms_per_step', 'ms_per_step_without_loss_item')})
