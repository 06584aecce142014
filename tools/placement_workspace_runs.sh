# three fresh processes each of the workspace-candidate and the output-candidate experiment; one line of analyze (workspaces) / fused mark (outputs) ms per process
for i in 1 2 3; do
  timeout -k 10 300 python tools/placement_workspace.py 8 > gpurun_out/r6_placement_workspace_run$i.txt 2>&1
  grep "workspace#" gpurun_out/r6_placement_workspace_run$i.txt | awk '{print $6}' | tr "\n" " "; echo
done
for i in 1 2 3; do
  timeout -k 10 300 python tools/placement_out.py 6 > gpurun_out/r6_placement_out_run$i.txt 2>&1
  grep "output#" gpurun_out/r6_placement_out_run$i.txt | awk '{print $6, $9}' | tr "\n" "|"; echo
done
